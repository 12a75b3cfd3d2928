cd $GRAFT_REPO_ROOT
L=variants/libekf_engine_trace.so
export EKF_PS_NOPRE=1
EKF_ENGINE_LIB=$L timeout 200 python scripts/persist_trace.py 1000 12 0 2>&1 | grep -v amdgpu.ids | head -n 34 | cut -c1-150
