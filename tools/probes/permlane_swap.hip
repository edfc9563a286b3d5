// what v_permlane16_swap / v_permlane32_swap do to two registers (gfx950): prints, per 16-lane row, which (register, row) each result row holds
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *out) {
    const unsigned l = threadIdx.x, a = 0x100 + l, b = 0x200 + l;
    const auto p = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    const auto q = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[l] = p[0]; out[64 + l] = p[1]; out[128 + l] = q[0]; out[192 + l] = q[1];
}
int main() {
    unsigned *d, h[256];
    hipMalloc(&d, sizeof(h));
    k<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char *names[4] = {"permlane16_swap(a, b)[0]", "permlane16_swap(a, b)[1]", "permlane32_swap(a, b)[0]", "permlane32_swap(a, b)[1]"};
    for (int r = 0; r < 4; ++r) {
        printf("%s:", names[r]);
        for (int row = 0; row < 4; ++row) printf("  row %d <- %c.row %u", row, (h[64 * r + 16 * row] >> 8) == 1 ? 'a' : 'b', (h[64 * r + 16 * row] & 0xff) / 16);
        printf("\n");
    }
    return 0;
}
