# usage (on the GPU box): bash tools/pmc_one.sh <tag> "<counters>" -- <command...>   -> gpurun_out/pmc_<tag>.txt (per-kernel means)
set -e
tag=$1; ctrs=$2; shift 3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/_pmc_$tag
rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d gpurun_out/_pmc_$tag -- "$@" > /dev/null 2>&1 || true
python - "$tag" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("gpurun_out/_pmc_%s/**/*counter_collection.csv" % tag, recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()
        tot[name][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(name, r["Counter_Name"])] += 1
with open("gpurun_out/pmc_%s.txt" % tag, "w") as out:
    for name in sorted(tot):
        out.write(name + "  " + "  ".join("%s=%.4g (n=%d)" % (c, v / cnt[(name, c)], cnt[(name, c)]) for c, v in sorted(tot[name].items())) + "\n")
PY
rm -rf gpurun_out/_pmc_$tag
