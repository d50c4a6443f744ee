mkdir -p gpurun_out/r6a
for k in 1 2 3 4; do python tools/diag/forward_determinism.py 600 24 "copy$k" > gpurun_out/r6a/det_$k.txt 2>&1 & done
SGRL_SET_GEMM=f32 python tools/diag/forward_determinism.py 600 24 "f32copy" > gpurun_out/r6a/det_f32.txt 2>&1 &
wait
cat gpurun_out/r6a/det_*.txt | grep -v amdgpu.ids
python tools/diag/forward_determinism.py 300 1024 "alone_big" 2>&1 | grep -v amdgpu.ids
