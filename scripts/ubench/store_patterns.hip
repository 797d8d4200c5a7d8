// How do 16-byte stores of a wave coalesce on gfx950?  A wave owns 64 voxels x 64 bytes (the cost volume's layout: 32 fp16 channels per
// voxel) and writes them with four global_store_dwordx4, in one of four lane -> address patterns (scripts/dev/store_patterns.py):
//   0  lane l, store j : voxel l, piece j            (64-byte stride between lanes: what a lane-owns-voxel kernel does naturally)
//   1  lane l, store j : byte 1024 j + 16 l          (fully contiguous)
//   2  lane (r = l >> 4, c = l & 15), store j : voxel 16 j + c, piece r    (the four pieces of a voxel in lanes c, c+16, c+32, c+48)
//   3  lane l, store j : voxel 16 j + (l >> 2), piece l & 3               (a quad of lanes writes one voxel = pattern 1's addresses)
// Build: hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o scripts/ubench/libsp.so scripts/ubench/store_patterns.hip; run: scripts/dev/store_patterns.py
// (round 4, MI355X, 251 MB: pattern 0 72.6 us; patterns 1, 2, 3 36.5-36.7 us -- profiles/r04_warp_lane_owner.txt).
// Pattern 0 also comes as "0s": the four stores spread over the kernel's run time instead of back to back.
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int PAT>
__global__ __launch_bounds__(256) void store_kernel(uint4* out, long n_wave_tiles, int spread) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long stride = (long)gridDim.x * 4;
    for (long t = wave; t < n_wave_tiles; t += stride) {
        uint4* base = out + t * 256;     // 64 voxels x 4 pieces of 16 bytes
        uint4 v = make_uint4((unsigned)t, lane, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            long idx;
            if (PAT == 0) idx = lane * 4 + j;
            else if (PAT == 1) idx = j * 64 + lane;
            else if (PAT == 2) idx = (16 * j + (lane & 15)) * 4 + (lane >> 4);
            else idx = (16 * j + (lane >> 2)) * 4 + (lane & 3);
            v.z = j;
            if (spread) {      // some arithmetic between the stores
                for (int k = 0; k < spread; ++k) v.w = v.w * 1664525u + 1013904223u;
            }
            base[idx] = v;
        }
    }
}

extern "C" int sp_run(int pat, void* out, long n_wave_tiles, int blocks, int spread, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    uint4* o = (uint4*)out;
    if (pat == 0) hipLaunchKernelGGL(store_kernel<0>, dim3(blocks), dim3(256), 0, st, o, n_wave_tiles, spread);
    else if (pat == 1) hipLaunchKernelGGL(store_kernel<1>, dim3(blocks), dim3(256), 0, st, o, n_wave_tiles, spread);
    else if (pat == 2) hipLaunchKernelGGL(store_kernel<2>, dim3(blocks), dim3(256), 0, st, o, n_wave_tiles, spread);
    else hipLaunchKernelGGL(store_kernel<3>, dim3(blocks), dim3(256), 0, st, o, n_wave_tiles, spread);
    return (int)hipGetLastError();
}
