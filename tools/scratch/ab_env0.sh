# same-box A/B of VAR=0 (feature off) vs unset (policy) on the bench: bash tools/scratch/ab_env0.sh VAR "bench args" [reps]
V=$1; ARGS=$2; N=${3:-3}
for i in $(seq $N); do for m in 0 p; do
  if [ $m = 0 ]; then export $V=0; else unset $V; fi
  python bench.py $ARGS --steps 300 --no-cpu-baseline --sustain-seconds 0 --no-secondary > gpurun_out/_ab.json
  python - <<PY
import json
r=json.load(open("gpurun_out/_ab.json"))
print("$V=$m", r["value"], r["ms_per_step"])
PY
done; done
