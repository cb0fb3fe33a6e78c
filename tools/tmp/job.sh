mkdir -p gpurun_out/x16
PS_XF_DEBUG=1 python tools/xf_probe.py c2 2>&1 | grep -v amdgpu.ids | tail -16 > gpurun_out/x16/c2.txt; cat gpurun_out/x16/c2.txt
