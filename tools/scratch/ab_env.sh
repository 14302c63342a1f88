# same-box A/B of an environment switch on the bench: bash tools/scratch/ab_env.sh VAR "bench args" [reps]  -> VAR=1 / unset alternately
V=$1; ARGS=$2; N=${3:-3}
for i in $(seq $N); do for m in 1 0; do
  if [ $m = 1 ]; then export $V=1; else unset $V; fi
  python bench.py $ARGS --steps 300 --no-cpu-baseline --sustain-seconds 0 --no-secondary > gpurun_out/_ab.json
  python - <<PY
import json
r=json.load(open("gpurun_out/_ab.json"))
print("$V=$m", r["value"], r["ms_per_step"])
PY
done; done
