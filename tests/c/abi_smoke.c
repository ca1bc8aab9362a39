/* abi_smoke.c -- the boundary from plain C: builds a small index through the C ABI only, searches it and prints the
 * result as text (the pytest harness compares it with the oracle).  Also the proof that include/ivfadc_hip.h is C.
 *   usage: abi_smoke <device> <seed>      (links against libivfadc_hip.so) */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "ivfadc_hip.h"

static unsigned long long s;
static float frand(void)
{
    s = s * 6364136223846793005ULL + 1442695040888963407ULL;
    return (float)((s >> 40) & 0xFFFF) / 65536.0f;
}

int main(int argc, char **argv)
{
    const int device = argc > 1 ? atoi(argv[1]) : 0;
    s = argc > 2 ? strtoull(argv[2], NULL, 10) : 1;
    enum { D = 12, KC = 7, M = 3, KSUB = 16, N = 90, NQ = 5, K = 4, W = 3 };
    static float cent[KC * D], cbs[D * KSUB], pts[N * D], qs[NQ * D], dists[NQ * K];
    static uint8_t labels[M * KSUB];
    static uint32_t ids[N], out_ids[NQ * K];
    static int32_t counts[NQ];
    int i;
    for (i = 0; i < KC * D; ++i) cent[i] = frand();
    for (i = 0; i < D * KSUB; ++i) cbs[i] = (frand() - 0.5f) * 0.4f;
    for (i = 0; i < M * KSUB; ++i) labels[i] = (uint8_t)(i % KSUB);
    for (i = 0; i < N * D; ++i) pts[i] = frand();
    for (i = 0; i < NQ * D; ++i) qs[i] = frand();
    for (i = 0; i < N; ++i) ids[i] = (uint32_t)i;

    if (ivfadc_abi_version() != IVFADC_ABI_VERSION) { fprintf(stderr, "library ABI %d, header ABI %d\n", ivfadc_abi_version(), IVFADC_ABI_VERSION); return 4; }
    ivfadc_t *h = NULL;
    if (ivfadc_create(&h, device, D, KC, M, KSUB, cent, cbs, labels) != IVFADC_OK) { fprintf(stderr, "create: %s\n", ivfadc_last_error()); return 2; }
    if (ivfadc_append(h, N, pts, ids, NULL, NULL) != IVFADC_OK) { fprintf(stderr, "append: %s\n", ivfadc_last_error()); return 2; }
    if (ivfadc_search(h, 0, qs, 0, 1, out_ids, dists, counts) != IVFADC_ERR_ASSERT) { fprintf(stderr, "k = 0 must assert\n"); return 3; }
    if (ivfadc_search(h, NQ, qs, K, W, out_ids, dists, counts) != IVFADC_OK) { fprintf(stderr, "search: %s\n", ivfadc_last_error()); return 2; }
    int64_t n = 0;
    ivfadc_ntotal(h, &n, NULL);
    printf("n %lld\n", (long long)n);
    for (i = 0; i < NQ; ++i) {
        int j;
        printf("q %d count %d:", i, (int)counts[i]);
        for (j = 0; j < counts[i]; ++j) printf(" %u %a", out_ids[i * K + j], dists[i * K + j]);
        printf("\n");
    }
    /* the arrays behind it, so the harness can rebuild the same index for the oracle */
    static int64_t offsets[KC + 1];
    static uint8_t codes[N * M];
    static uint32_t lids[N];
    if (ivfadc_get_lists(h, offsets, codes, lids) != IVFADC_OK) return 2;
    printf("offsets");
    for (i = 0; i <= KC; ++i) printf(" %lld", (long long)offsets[i]);
    printf("\ncodes");
    for (i = 0; i < N * M; ++i) printf(" %u", codes[i]);
    printf("\nids");
    for (i = 0; i < N; ++i) printf(" %u", lids[i]);
    printf("\ncent");
    for (i = 0; i < KC * D; ++i) printf(" %a", cent[i]);
    printf("\ncbs");
    for (i = 0; i < D * KSUB; ++i) printf(" %a", cbs[i]);
    printf("\nqs");
    for (i = 0; i < NQ * D; ++i) printf(" %a", qs[i]);
    printf("\n");
    ivfadc_destroy(h);
    return 0;
}
