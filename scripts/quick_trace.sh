cd $GRAFT_REPO_ROOT
L=variants/libekf_engine_trace.so
EKF_ENGINE_LIB=$L timeout 200 python scripts/persist_trace.py 1000 12 0 2>&1 | grep -v amdgpu.ids > gpurun_out/trace_li.txt
EKF_ENGINE_LIB=$L timeout 200 python scripts/persist_trace.py 1000 12 1 2>&1 | grep -v amdgpu.ids > gpurun_out/trace_hi.txt
EKF_ENGINE_LIB=$L PRECISION=0 timeout 200 python scripts/persist_trace.py 200 12 0 2>&1 | grep -v amdgpu.ids > gpurun_out/trace_n200.txt
