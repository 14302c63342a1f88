# A/B of the optimizer launch that adds up the backward's last wgrad slices itself (lidbox_adam_step_jobs) against the
# stand-alone slice-sum launch + adam_kernel (LIDBOX_ADAM_NO_FOLD=1), interleaved on ONE box, fp32 and bf16 at bs 256.
# usage: gpurun -- 'bash tools/ab_adam_fold.sh'
R='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["ms_per_step"], d["value"])'
F="--no-cpu-baseline --no-secondary --no-kernel-timing --steps 200 --warmup 20"
for i in 1 2 3; do
python bench.py $F | python -c "$R" "fp32 fold"
LIDBOX_ADAM_NO_FOLD=1 python bench.py $F | python -c "$R" "fp32 launch"
python bench.py --compute-dtype bfloat16 $F | python -c "$R" "bf16 fold"
LIDBOX_ADAM_NO_FOLD=1 python bench.py --compute-dtype bfloat16 $F | python -c "$R" "bf16 launch"
done
