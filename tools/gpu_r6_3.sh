for v in base wg_rot5 wg_odd1 wg_rot5odd1 wg_rot13; do
  if [ $v = base ]; then L="X=1"; else L="UNCR_HIP_LIB=$PWD/uncrtaints_amd/lib/ablate/lib_$v.so"; fi
  echo "== $v"; env $L python tools/time_wgrad.py 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r06_time_wgrad.log
REPS=2 bash tools/ab_variants.sh wg_rot5 wg_odd1 wg_rot5odd1 wg_rot13 2>&1 | tee gpurun_out/r06_ab_wgrot.log
