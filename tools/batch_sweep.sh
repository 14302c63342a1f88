# usage (MI355X box): bash tools/batch_sweep.sh [extra bench args] -> utterances/s and ms/step of the train step per batch size
for B in 32 64 128 256 512 1024; do
  python bench.py --batch $B --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-secondary "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%5d  %9.0f  %8.4f' % ($B, d['value'], d['ms_per_step']))"
done
