for rep in 1 2; do
for v in base wg_rot1 wg_rot3 wg_rot5 wg_rot7 wg_rot11 wg_rot99 wg_rot5odd1; do
  if [ $v = base ]; then L="X=1"; else L="UNCR_HIP_LIB=$PWD/uncrtaints_amd/lib/ablate/lib_$v.so"; fi
  echo "== $v"; env $L python tools/time_wgrad.py 2>&1 | grep -v amdgpu.ids
done; done | tee gpurun_out/r06_time_wgrad2.log
run() {
  if [ $1 = base ]; then L="X=1"; else L="UNCR_HIP_LIB=$PWD/uncrtaints_amd/lib/ablate/lib_$1.so"; fi
  env $L python bench.py --no-cpu-baseline --no-bf16-leg --no-kernel-events --steps 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"
}
run base > /dev/null
for i in 1 2 3; do for v in wg_rot5 base wg_rot7 base wg_rot99 base; do run $v; done; done | tee gpurun_out/r06_ab_wgrot2.log
