# Interleaved A/B of the train step (fp32 and bf16, bs 256) on ONE box: the in-tree library against tools/ab_ship/libprev.so.
# Make libprev.so from the previous sources first (here, not on the GPU box):
#   git stash; LIDBOX_AB_DIR=ab_ship python tools/ab_build.py prev gemm.hip; git stash pop; python -c "import __graft_entry__ as g; g.build()"
# then: gpurun -- 'bash tools/ab_step.sh'      (tools/ab_ship/ travels to the box, tools/ab/ does not: .gpurunignore)
for i in 1 2; do
python bench.py --no-cpu-baseline --no-secondary --no-kernel-timing --steps 200 --warmup 20 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new', d['ms_per_step'], d['value'])"
LIDBOX_HIP_LIB=tools/ab_ship/libprev.so python bench.py --no-cpu-baseline --no-secondary --no-kernel-timing --steps 200 --warmup 20 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('prev', d['ms_per_step'], d['value'])"
python bench.py --compute-dtype bfloat16 --no-cpu-baseline --no-secondary --no-kernel-timing --steps 200 --warmup 20 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bf16 new', d['ms_per_step'], d['value'])"
LIDBOX_HIP_LIB=tools/ab_ship/libprev.so python bench.py --compute-dtype bfloat16 --no-cpu-baseline --no-secondary --no-kernel-timing --steps 200 --warmup 20 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bf16 prev', d['ms_per_step'], d['value'])"
done
