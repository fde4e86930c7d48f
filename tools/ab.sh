# A/B of two builds of the library on the headline bench (alternating runs): bash tools/ab.sh <other.so> <n>
other=$1; n=${2:-3}
for k in $(seq 1 $n); do
  timeout -k 10 100 python bench.py --no-cpu-baseline --no-secondary > gpurun_out/ab_new_$k.json 2>/dev/null || exit 1
  SLAM_HIP_LIB=$other timeout -k 10 100 python bench.py --no-cpu-baseline --no-secondary > gpurun_out/ab_old_$k.json 2>/dev/null || exit 1
done
