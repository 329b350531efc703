export N=10 ROUNDS=9
V="old=old;mix=;mix_nostag=:stagger=2;lean=lean;lean_nostag=lean:stagger=2;padded=padded;padded_nostag=padded:stagger=2"
for shape in "16 256 256 128 128 9 2" "16 256 256 256 128 9 2" "32 256 256 128 128 9 1"; do echo "== conv $shape"; VARIANTS="$V" python tools/ab.py conv $shape 2>&1 | grep -v amdgpu.ids; done
V="old=old;mix=;mix_stag=:stagger=1;padded=padded;padded_stag=padded:stagger=1"
for shape in "16 128 128 256 256 9 2" "16 64 64 384 384 9 2"; do echo "== conv $shape"; VARIANTS="$V" python tools/ab.py conv $shape 2>&1 | grep -v amdgpu.ids; done
