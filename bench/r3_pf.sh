set -e
echo "== conv3_1 forward (128 -> 256 @ 56, batch 1024), cfg 18"
N=1024 HW=56 C=128 CO=256 CFG=18 timeout -k 10 120 python bench/stamp_igemm.py 2>&1 | grep -v amdgpu.ids
echo "== conv3_2 forward (256 -> 256 @ 56, batch 1024), cfg 18"
N=1024 HW=56 C=256 CO=256 CFG=18 timeout -k 10 120 python bench/stamp_igemm.py 2>&1 | grep -v amdgpu.ids
echo "== conv2_2 forward (128 -> 128 @ 112, batch 512), cfg 19"
N=512 HW=112 C=128 CO=128 CFG=19 timeout -k 10 120 python bench/stamp_igemm.py 2>&1 | grep -v amdgpu.ids
