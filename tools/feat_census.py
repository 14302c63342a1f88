"""ISA census of feat512_stream_kernel<2, true, false, false, 13> (the log-mel instantiation of the train step) from hipcc's device assembly.

    python tools/feat_census.py [--timing]      (build container or GPU box: needs hipcc only)

Compiles csrc/features.hip to gfx950 assembly, cuts the kernel into basic blocks, and counts instructions by class per
block.  With --timing the kernel is built with -DLBX_FEAT_TIMING=0, whose s_memtime stamps mark the phase boundaries of a
tile (load+window | pass-1 DFT | twiddle | exchange | pass-2 DFT | untangle | mel | store): the counts between two stamps
are the phase's instructions (the stamped build's code is the production code plus the stamps).
Static counts of straight-line blocks; a block inside a loop is annotated with its label so that trip counts can be applied
by hand (the mel run loop: ceil(seg_len / 4) trips; the shuffle-combine loop: seg_steps trips)."""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
timing = "--timing" in sys.argv
out = "/tmp/feat_census%s.s" % ("_t" if timing else "")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
       "-I" + os.path.join(ROOT, "lidbox_amd", "csrc"), "--cuda-device-only", "-S", os.path.join(ROOT, "lidbox_amd", "csrc", "features.hip"), "-o", out]
if timing:
    cmd.insert(1, "-DLBX_FEAT_TIMING=0")
subprocess.run(cmd, check=True, capture_output=True)
NW = next((a.split("=")[1] for a in sys.argv if a.startswith("--nw=")), "4")        # --nw=14: the wide workgroup's instantiation
KERNEL = "_ZN12_GLOBAL__N_121feat512_stream_kernelILi2ELb1ELb0ELb0ELi13EEEvNS_9FusedArgsE"
lines = open(out).read().split("\n")
i0 = lines.index(KERNEL + ": ; @" + KERNEL) if (KERNEL + ": ; @" + KERNEL) in lines else next(i for i, l in enumerate(lines) if l.startswith(KERNEL + ":"))
i1 = next(i for i in range(i0, len(lines)) if lines[i].strip().startswith("s_endpgm"))

def klass(op):
    if op.startswith("v_pk_"): return "valu_packed"
    if op.startswith("v_mfma") or op.startswith("v_smfma"): return "mfma"
    if op in ("v_log_f32", "v_exp_f32", "v_rcp_f32", "v_sqrt_f32", "v_rsq_f32", "v_sin_f32", "v_cos_f32"): return "valu_trans"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")): return "valu_lane"
    if op.startswith("v_cndmask"): return "valu_select"
    if op.startswith(("v_mov", "v_accvgpr")): return "valu_mov"
    if op.startswith(("v_fma_f32", "v_fmac_f32", "v_mad_f32")): return "valu_fma"
    if op.startswith(("v_add_f32", "v_sub_f32", "v_subrev_f32")): return "valu_addsub"
    if op.startswith("v_mul_f32"): return "valu_mul"
    if op.startswith(("v_cmp", "v_cmpx")): return "valu_cmp"
    if op.startswith("v_"): return "valu_int_other"
    if op.startswith("ds_bpermute") or op.startswith("ds_swizzle") or op.startswith("ds_permute"): return "lds_shuffle"
    if op.startswith("ds_read") or op.startswith("ds_load"): return "lds_read"
    if op.startswith("ds_write") or op.startswith("ds_store"): return "lds_write"
    if op.startswith(("global_load", "buffer_load", "flat_load")): return "vmem_load"
    if op.startswith(("global_store", "buffer_store", "flat_store")): return "vmem_store"
    if op.startswith(("scratch_",)): return "scratch"
    if op == "s_waitcnt": return "s_waitcnt"
    if op.startswith("s_memtime"): return "STAMP"
    if op.startswith(("s_cbranch", "s_branch")): return "branch"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_nop"): return "s_nop"
    if op.startswith("s_"): return "salu"
    return "other"

blocks, cur, name = [], collections.Counter(), "entry"
order = []
for l in lines[i0 + 1:i1 + 1]:
    t = l.strip()
    if not t or t.startswith((";", ".")) and not re.match(r"^\.LBB\d+_\d+:", t):
        continue
    m = re.match(r"^(\.LBB\d+_\d+):", t)
    if m:
        blocks.append((name, cur)); cur, name = collections.Counter(), m.group(1)
        continue
    op = t.split()[0]
    k = klass(op)
    if k == "STAMP":
        blocks.append((name, cur)); cur, name = collections.Counter(), name + "+stamp"
        continue
    cur[k] += 1
blocks.append((name, cur))
VALU = [k for k in ("valu_fma", "valu_addsub", "valu_mul", "valu_packed", "valu_trans", "valu_select", "valu_mov", "valu_cmp", "valu_lane", "valu_int_other")]
cols = VALU + ["lds_read", "lds_write", "lds_shuffle", "vmem_load", "vmem_store", "salu", "s_waitcnt", "s_nop", "branch", "scratch"]
print("%-22s %6s | " % ("block", "VALU") + " ".join("%7s" % c.replace("valu_", "")[:7] for c in cols))
tot = collections.Counter()
for name, c in blocks:
    n = sum(c.values())
    if n == 0:
        continue
    v = sum(c[k] for k in VALU)
    tot.update(c)
    if n >= 12:
        print("%-22s %6d | " % (name[:22], v) + " ".join("%7d" % c[k] for k in cols))
print("%-22s %6d | " % ("whole kernel (static)", sum(tot[k] for k in VALU)) + " ".join("%7d" % tot[k] for k in cols))
