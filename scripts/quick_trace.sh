cd $GRAFT_REPO_ROOT
L=variants/libekf_engine_trace.so
EKF_ENGINE_LIB=$L timeout 200 python scripts/persist_trace.py 1000 12 0 2>&1 | grep -v amdgpu.ids | head -n 32 | cut -c1-125
EKF_ENGINE_LIB=$L timeout 200 python scripts/persist_trace.py 1000 12 1 2>&1 | grep -v amdgpu.ids | head -n 16 | cut -c1-125
