python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python tools/esl_probe.py 2>&1 | grep -v amdgpu.ids | tail -2
STRIDE=3 python tools/esl_probe.py 2>&1 | grep -v amdgpu.ids | tail -2
STRIDE=3 XM_K1_DIRECT=1 python tools/esl_probe.py 2>&1 | grep -v amdgpu.ids | tail -1
STRIDE=1 python tools/esl_probe.py 2>&1 | grep -v amdgpu.ids | tail -2
STRIDE=1 XM_K1_DIRECT=1 python tools/esl_probe.py 2>&1 | grep -v amdgpu.ids | tail -1
python tools/scale_probe.py 2>&1 | grep -v amdgpu | head -4
