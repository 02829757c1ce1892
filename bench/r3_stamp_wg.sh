set -e
mkdir -p gpurun_out/r3stamp
for shape in "256 56 256 256" "256 28 512 512" "256 112 128 128"; do
set -- $shape
N=$1 HW=$2 C=$3 CO=$4 DTP_HIP_LIB=bench/stamp_libdtp_hip.so timeout -k 10 120 python bench/stamp_wgrad2.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r3stamp/wgrad2_stamps.txt
done
