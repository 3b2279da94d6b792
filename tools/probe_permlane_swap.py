import torch, os, subprocess, ctypes
src = r'''
#include <hip/hip_runtime.h>
typedef unsigned u2 __attribute__((ext_vector_type(2)));
extern "C" __global__ void k(int* o) {
    unsigned a = threadIdx.x, b = 100 + threadIdx.x;
    u2 r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    o[threadIdx.x] = r.x; o[64 + threadIdx.x] = r.y;
    u2 q = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    o[128 + threadIdx.x] = q.x; o[192 + threadIdx.x] = q.y;
    unsigned v = threadIdx.x;
    u2 s = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    o[256 + threadIdx.x] = s.x; o[320 + threadIdx.x] = s.y;
}
extern "C" void run(int* o) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o); hipDeviceSynchronize(); }
'''
open('/tmp/ps.hip','w').write(src)
subprocess.check_call(['/opt/rocm/bin/hipcc','--offload-arch=gfx950','-O3','-shared','-fPIC','-o','/tmp/ps.so','/tmp/ps.hip'])
lib = ctypes.CDLL('/tmp/ps.so')
o = torch.zeros(384, dtype=torch.int32, device='cuda')
lib.run(ctypes.c_void_p(o.data_ptr()))
o = o.cpu().view(6, 64)
names = ['p16 r.x (a=lane, b=100+lane)', 'p16 r.y', 'p32 q.x', 'p32 q.y', 'p16 same-operand s.x', 'p16 same s.y']
for n, row in zip(names, o): print(n, row.tolist())
