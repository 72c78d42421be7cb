run() { echo "== $1"; env $1 bash scripts/gpu_quick.sh envs 2>&1 | tail -2 | cut -c1-190; }
run "COOK_DUMMY=1"
run "HSA_ENABLE_INTERRUPT=0"
run "ROC_ACTIVE_WAIT_TIMEOUT=2000"
run "HSA_ENABLE_INTERRUPT=0 ROC_ACTIVE_WAIT_TIMEOUT=2000"
