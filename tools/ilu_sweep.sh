cd $GRAFT_REPO_ROOT
for bal in cells owned; do for bpw in 1 2; do for wide in 1; do
  NSX_DEBUG=1 NSX_BPW_F=$bpw python bench.py --no-cpu --steps 3 --warmup 1 --spinup 0 --profile-steps 2 --balance $bal > gpurun_out/sw_${bal}_${bpw}.json 2> gpurun_out/sw_${bal}_${bpw}.err
  python3 - <<PY
import json
p=json.loads(open('gpurun_out/sw_${bal}_${bpw}.json').read())
k=p['kernels']
print('$bal bpw=$bpw', 'ilu_solve_F %.2f us  spmv_F %.2f  mgs %.2f  ms/outer %.3f'%(k['ilu_solve_F']['avg_us'],k['spmv_F']['avg_us'],k['mgs_sweep']['avg_us'],p['ms_per_outer_iteration']), 'F its/outer %.1f'%(p['inner_F_iters_per_step']/p['gmres_outer_iters_per_step']))
PY
  grep "ilu schedule: rows 347886" gpurun_out/sw_${bal}_${bpw}.err
done; done; done
