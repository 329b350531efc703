python -m pytest tests/test_hip_ops.py -m gpu -q -p no:cacheprovider -k "conv" 2>&1 | tail -2
python tools/clock_probe.py conv 16 256 256 128 128 2>&1 | grep -v amdgpu.ids
python tools/clock_probe.py conv 16 128 128 256 256 2>&1 | grep -v amdgpu.ids
export N=10 ROUNDS=7
for shape in "16 256 256 128 128 9 2" "16 128 128 256 256 9 2" "16 64 64 384 384 9 2" "32 128 128 256 768 1 0"; do echo "== conv $shape"; VARIANTS="old=old;new=" python tools/ab.py conv $shape 2>&1 | grep -v amdgpu.ids; done
