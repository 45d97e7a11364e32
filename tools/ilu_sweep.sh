cd $GRAFT_REPO_ROOT
for bal in cells owned; do for r in 90 130; do
  if [ $r = 128x ]; then export NSX_SPMV_BY_RANK=0; else export NSX_SPMV_BY_RANK=1; export NSX_SPMV_R=$r; fi
  NSX_DEBUG=1 python bench.py --no-cpu --steps 3 --warmup 1 --spinup 0 --profile-steps 2 --balance $bal > gpurun_out/sw_${bal}_${r}.json 2> gpurun_out/sw_${bal}_${r}.err
  python3 - <<PY
import json
p=json.loads(open('gpurun_out/sw_${bal}_${r}.json').read())
k=p['kernels']
print('$bal R=$r', 'ilu_solve_F %.2f us  spmv_F %.2f  mgs %.2f  ms/outer %.3f'%(k['ilu_solve_F']['avg_us'],k['spmv_F']['avg_us'],k['mgs_sweep']['avg_us'],p['ms_per_outer_iteration']), 'F its/outer %.1f'%(p['inner_F_iters_per_step']/p['gmres_outer_iters_per_step']))
PY
  grep "blocked spmv" gpurun_out/sw_${bal}_${r}.err
done; done
