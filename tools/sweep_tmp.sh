python -m pytest tests -x -q -m gpu 2>&1 | tail -3
run() { python bench.py --no-cpu-baseline "$@" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'],d['ms_per_step'],d['roofline']['avg_launch_us'], d['roofline']['frame_us_serial'])"; }
echo "general"; run; run --slots 1
echo "assume-sorted"; run --assume-sorted; run --assume-sorted --slots 1; run --assume-sorted --slots 16
