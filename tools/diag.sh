cd $GRAFT_REPO_ROOT
for v in NO_BARRIER NO_SHUFFLE; do
  cp poroelasticity_dealii_amd/lib/libporoel_hip.so /tmp/keep.so
  cp poroelasticity_dealii_amd/lib/libporoel_hip_$v.so poroelasticity_dealii_amd/lib/libporoel_hip.so
  echo $v; PORO_DIAG_SKIP_SELFCHECK=1 timeout -k 10 120 python tools/bench_ops.py 3,72,2,mf 2>&1 | tail -1 | cut -c1-140
  cp /tmp/keep.so poroelasticity_dealii_amd/lib/libporoel_hip.so
done
