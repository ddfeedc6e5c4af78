// The 1-D F(4x4,3x3) transforms shared by the Winograd kernels (winograd.hip: the fused kernels; winogemm.hip: the packed input
// transform and the contraction kernel's output epilogue).  Interpolation points 0, +-1, +-2, inf (Lavin & Gray):
//   B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
//   A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
#pragma once
#include <hip/hip_runtime.h>

namespace w2e {

template <class T>
__device__ __forceinline__ void wino4_bt(const T (&d)[6], T (&t)[6]) {  // t = B^T d  (T: float, or a float2 vector -> v_pk_*_f32)
    const T a = d[4] - 4.f * d[2], b = d[3] - 4.f * d[1], c = d[4] - d[2], e = 2.f * (d[3] - d[1]);
    t[0] = 4.f * d[0] - 5.f * d[2] + d[4];
    t[1] = a + b;
    t[2] = a - b;
    t[3] = c + e;
    t[4] = c - e;
    t[5] = 4.f * d[1] - 5.f * d[3] + d[5];
}

template <class T>
__device__ __forceinline__ void wino4_at(const T (&m)[6], T (&y)[4]) {  // y = A^T m
    const T p = m[1] + m[2], q = m[1] - m[2], r = m[3] + m[4], s = m[3] - m[4];
    y[0] = m[0] + p + r;
    y[1] = q + 2.f * s;
    y[2] = p + 4.f * r;
    y[3] = q + 8.f * s + m[5];
}

}  // namespace w2e
