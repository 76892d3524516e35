/* asan_check.c -- TEST INFRASTRUCTURE ONLY: runs the CPU restatement under AddressSanitizer + UBSan on shapes that
 * exercise every path-geometry case (wide, tall, square, one row / one column, census no-op sizes).
 * Built and run by tests/test_oracle_sanitizers.py:  gcc -fsanitize=address,undefined sgm_oracle.c asan_check.c */
#include "sgm_oracle.h"
#include <stdio.h>
#include <stdlib.h>

int main(void)
{
    static const int shapes[][4] = {{24, 16, 0, 8},  {20, 31, 0, 8}, {33, 33, 2, 14}, {1, 9, 0, 2}, {9, 1, 0, 2},
                                    {2, 2, 0, 1},    {5, 30, 0, 4},  {64, 3, 0, 8},   {70, 33, 0, 16}, {130, 40, 5, 69}};
    sgmo_ctx* c = sgmo_create();
    unsigned long long checksum = 0;
    for (unsigned i = 0; i < sizeof shapes / sizeof shapes[0]; ++i) {
        const int W = shapes[i][0], H = shapes[i][1], dmin = shapes[i][2], dmax = shapes[i][3];
        uint8_t* l = malloc((size_t)W * H);
        uint8_t* r = malloc((size_t)W * H);
        float* d = malloc(sizeof(float) * (size_t)W * H);
        sgmo_synth_pair(W, H, dmax - dmin, 0xA5A50000u + i, l, r);
        sgmo_option o = {8, (uint16_t)dmin, (uint16_t)dmax, true, 0.99f, true, 1.0f, true, 6, 10, 150};
        if (!sgmo_reset(c, (uint16_t)W, (uint16_t)H, &o) || !sgmo_match(c, l, r, d)) { fprintf(stderr, "match failed\n"); return 1; }
        for (int k = 0; k < W * H; ++k) checksum += (d[k] == d[k] && d[k] < 1e30f) ? (unsigned long long)(d[k] * 16.0f) : 7;
        free(l); free(r); free(d);
    }
    sgmo_destroy(c);
    printf("asan_check ok %llu\n", checksum);
    return 0;
}
