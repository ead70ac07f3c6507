// What ds_read_b64_tr_b8 (gfx950) returns: LDS holds lds[i] = i (16-bit tags written as two byte planes), every lane passes its own
// address; the 8 bytes each lane receives are printed for a few address patterns.  hipcc --offload-arch=gfx950 -O2 tr8_probe.hip -o tr8_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v2i __attribute__((ext_vector_type(2)));
__global__ void probe(int pattern, unsigned char* out_lo, unsigned char* out_hi) {
    __shared__ unsigned char lo[2048], hi[2048];
    for (int i = threadIdx.x; i < 2048; i += 64) { lo[i] = (unsigned char)(i & 255); hi[i] = (unsigned char)(i >> 8); }
    __syncthreads();
    const int l = threadIdx.x;
    int addr = 0;
    if (pattern == 0) addr = l * 8;                                  // lane-linear
    if (pattern == 1) addr = (l & 15) * 16 + (l >> 4) * 256;         // one 16-byte row per lane, first 8 bytes
    if (pattern == 2) addr = (l & 15) * 16 + 8 + (l >> 4) * 256;     // ... second 8 bytes
    if (pattern == 3) addr = (l >> 1 & 7) * 16 + (l & 1) * 8 + (l >> 4) * 128;   // [8][16] block per 16 lanes, lane i -> row i/2, half i%2
    v2i a = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) v2i*)(lo + addr));
    v2i b = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) v2i*)(hi + addr));
    ((v2i*)out_lo)[l] = a; ((v2i*)out_hi)[l] = b;
}
int main() {
    unsigned char *dl, *dh, hl[512], hh[512];
    hipMalloc(&dl, 512); hipMalloc(&dh, 512);
    for (int p = 0; p < 4; ++p) {
        probe<<<1, 64>>>(p, dl, dh);
        hipMemcpy(hl, dl, 512, hipMemcpyDeviceToHost); hipMemcpy(hh, dh, 512, hipMemcpyDeviceToHost);
        printf("pattern %d\n", p);
        for (int l = 0; l < 64; ++l) {
            printf(" lane %2d:", l);
            for (int j = 0; j < 8; ++j) printf(" %4d", hl[l * 8 + j] + 256 * hh[l * 8 + j]);
            printf("\n");
        }
    }
    return 0;
}
