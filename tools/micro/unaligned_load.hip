#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const unsigned char *b, unsigned *out, int n) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = *reinterpret_cast<const unsigned *>(b + 3 * (size_t)i) & 0xFFFFFFu;
}
int main() {
    const int n = 1 << 20;
    std::vector<unsigned char> h(3 * (size_t)n + 8);
    for (size_t i = 0; i < h.size(); i++) h[i] = (unsigned char)(i * 131u + (i >> 8) * 7u);
    unsigned char *d; unsigned *o;
    hipMalloc(&d, h.size()); hipMalloc(&o, n * 4);
    hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, o, n);
    std::vector<unsigned> r(n);
    hipError_t e = hipMemcpy(r.data(), o, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; i++) { unsigned w = h[3 * (size_t)i] | (h[3 * (size_t)i + 1] << 8) | (h[3 * (size_t)i + 2] << 16); if (w != r[i]) bad++; }
    printf("unaligned dword loads: err %d mismatches %d of %d\n", (int)e, bad, n);
    return bad != 0;
}
