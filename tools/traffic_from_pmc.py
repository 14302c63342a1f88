"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), per launch.
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 128-B requests as 64 B for wide
coalesced reads -> doubled; both counters are in KiB.  usage: traffic_from_pmc.py <fetch_dir> <write_dir> <out.json>"""
import csv, glob, json, os, re, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

def load(d, counter):
    tot, cnt = collections.defaultdict(float), collections.Counter()
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            name = name.split("(")[0].strip()
            tot[name] += float(r["Counter_Value"]); cnt[name] += 1
    return {k: tot[k] / cnt[k] for k in tot}, cnt

fetch, n = load(sys.argv[1], "FETCH_SIZE")
write, _ = load(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, 0.0), write.get(k, 0.0)
    out[k] = {"launches": n.get(k, 0), "fetch_kib_raw": round(f, 1), "write_kib": round(w, 1),
              "hbm_bytes_per_launch": int((2.0 * f + w) * 1024)}
json.dump({"source_hash": bench.source_hash(), "note": "hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024; FETCH_SIZE doubled per the gfx950 "
                   "calibration in MI355X_MICROARCH.md; averaged over all launches of the kernel in `python bench.py "
                   "--steps 10 --warmup 3 --no-cpu-baseline`; source_hash = sha1 of csrc/*.hip, *.h and include/*.h the counters were taken "
                   "from (bench.py quotes a summary only next to the same sources)", "kernels": out}, open(sys.argv[3], "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:14]:
    print("%-52s launches=%-5d %8.1f MB/launch" % (k[:52], v["launches"], v["hbm_bytes_per_launch"] / 1e6))
