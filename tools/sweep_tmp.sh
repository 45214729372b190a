python -m pytest tests -x -q -m gpu 2>&1 | tail -3
run() { python bench.py --no-cpu-baseline "$@" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'],d['ms_per_step'],d['roofline']['avg_launch_us'])"; }
echo "default slots4"; run
echo "slots 8"; run --slots 8
echo "slots 1"; run --slots 1
echo "w 4/8 slots 8"; XM_W_TS=4 XM_W_X=8 run --slots 8
echo "graph 64 slots 8"; run --slots 8 --graph --steps 64
