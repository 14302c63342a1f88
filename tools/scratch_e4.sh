for i in 1 2; do
python bench.py --no-cpu-baseline --no-secondary --no-kernel-timing --steps 200 --warmup 20 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new', d['ms_per_step'], d['value'])"
LIDBOX_HIP_LIB=tools/ab_ship/libprev.so python bench.py --no-cpu-baseline --no-secondary --no-kernel-timing --steps 200 --warmup 20 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('prev', d['ms_per_step'], d['value'])"
python bench.py --compute-dtype bfloat16 --no-cpu-baseline --no-secondary --no-kernel-timing --steps 200 --warmup 20 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bf16 new', d['ms_per_step'], d['value'])"
LIDBOX_HIP_LIB=tools/ab_ship/libprev.so python bench.py --compute-dtype bfloat16 --no-cpu-baseline --no-secondary --no-kernel-timing --steps 200 --warmup 20 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bf16 prev', d['ms_per_step'], d['value'])"
done
