# A/B of one environment switch on the same box: rocprofv3 kernel stats of the bench for each value
# usage: bash tools/ab_prof.sh VAR v1 v2 ...   (summaries land in gpurun_out/ab_<VAR>_<value>/)
# Two BUILDS are compared with VAR=AVA_HIP_LIB_TAG: value x loads csrc/libava_hip_x.so, value "main" the default library.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
VAR=$1; shift
for v in "$@"; do
  if [ "$VAR" = AVA_HIP_LIB_TAG ] && [ "$v" = main ]; then unset AVA_HIP_LIB_TAG; else export $VAR=$v; fi
  rocprofv3 --kernel-trace --stats -d gpurun_out/ab_${VAR}_$v -o ab --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab_${VAR}_$v.json 2> gpurun_out/ab_${VAR}_$v.err
done
