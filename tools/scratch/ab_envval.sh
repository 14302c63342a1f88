# same-box A/B of VAR=VALUE vs unset on the bench: bash tools/scratch/ab_envval.sh VAR VALUE "bench args" [reps]
V=$1; VAL=$2; ARGS=$3; N=${4:-3}
for i in $(seq $N); do for m in 1 0; do
  if [ $m = 1 ]; then export $V="$VAL"; else unset $V; fi
  python bench.py $ARGS --steps 300 --no-cpu-baseline --sustain-seconds 0 --no-secondary > gpurun_out/_ab.json
  python - <<PY
import json
r=json.load(open("gpurun_out/_ab.json"))
print("$V=$m", r["value"], r["ms_per_step"])
PY
done; done
