mkdir -p gpurun_out/r6a /tmp/swap
python tools/diag/round1_data_swap.py collect /tmp/swap/d_f16x3.pt 2>&1 | grep -v amdgpu.ids &
SGRL_SET_GEMM=f32 python tools/diag/round1_data_swap.py collect /tmp/swap/d_f32.pt 2>&1 | grep -v amdgpu.ids &
wait
for data in f16x3 f32; do
  python tools/diag/round1_data_swap.py update /tmp/swap/d_$data.pt 2>&1 | grep -v amdgpu.ids &
  SGRL_SET_GEMM=f32 python tools/diag/round1_data_swap.py update /tmp/swap/d_$data.pt 2>&1 | grep -v amdgpu.ids &
done
wait
