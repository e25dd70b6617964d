cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -m gpu -q -x > gpurun_out/h_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/h_pytest.log
tail -12 gpurun_out/h_pytest.log
for i in 1 2; do timeout 900 python bench.py --no-cpu-baseline --steps 100 --warmup 30 --global-batch 0 --no-loader-path > gpurun_out/h_bench128_$i.json 2> gpurun_out/h_bench128.err; python -c "
import json; d=json.load(open('gpurun_out/h_bench128_$i.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['ms_per_step_by_category'], d['roofline']['launch_groups_per_step'])"; done
