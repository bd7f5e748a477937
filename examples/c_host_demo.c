/* A plain-C host of libcapdec_hip.so: no Python, no torch, nothing but include/capdec.h.
 *
 *   gcc -O2 -Iinclude examples/c_host_demo.c -Lcapdec_amd/lib -lcapdec_hip -Wl,-rpath,$PWD/capdec_amd/lib -o /tmp/c_host_demo
 *   /tmp/c_host_demo model.bin [entry_length] [beam]
 *   /tmp/c_host_demo model.bin T beam RANK NRANKS IDFILE      one process per GPU (device = RANK): the captions are
 *       sharded with capdec_shard_bounds, rank 0 creates the RCCL id (capdec_comm_unique_id) and publishes it through
 *       IDFILE, every rank joins with capdec_comm_init and the generated ids are collected with capdec_gather_ids --
 *       the whole multi-GPU path of SURVEY section 8 row E without Python or torch.distributed.
 *
 * model.bin (little endian, written by tests/test_hip_parity.py::test_c_host_without_torch or any exporter):
 *   int32 magic 0x43415044 ("CAPD"), n_layer, n_head, n_embd, vocab, n_pos; float32 ln_eps;
 *   float32 arrays in state-dict order: wte, wpe, per layer {ln_1.w, ln_1.b, c_attn.w [d,3d], c_attn.b, c_proj.w [d,d],
 *   c_proj.b, ln_2.w, ln_2.b, c_fc.w [d,4d], c_fc.b, mlp.c_proj.w [4d,d], mlp.c_proj.b}, ln_f.w, ln_f.b;
 *   int32 n, P; float32 prefix embeddings [n, P, d]   (what `model.clip_project(prefix)` produced)
 * Output (stdout): one line per caption "greedy <len> : ids..." then "beam <len> <mean-logprob> : ids..." (best beam). */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "capdec.h"

#define CHECK(call)                                                                        \
    do {                                                                                   \
        if ((call) != 0) {                                                                 \
            fprintf(stderr, "%s failed: %s\n", #call, capdec_last_error());                \
            return 1;                                                                      \
        }                                                                                  \
    } while (0)

static float *read_floats(FILE *f, size_t n) {
    float *p = (float *)malloc(n * sizeof(float));
    if (!p || fread(p, sizeof(float), n, f) != n) {
        fprintf(stderr, "short read (%zu floats)\n", n);
        exit(2);
    }
    return p;
}

int main(int argc, char **argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: %s model.bin [entry_length] [beam]\n", argv[0]);
        return 2;
    }
    const int T = argc > 2 ? atoi(argv[2]) : 12, beam = argc > 3 ? atoi(argv[3]) : 5;
    const int rank = argc > 6 ? atoi(argv[4]) : 0, nranks = argc > 6 ? atoi(argv[5]) : 1;
    const char *idfile = argc > 6 ? argv[6] : NULL;
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    int32_t hdr[6];
    float eps;
    if (fread(hdr, 4, 6, f) != 6 || hdr[0] != 0x43415044 || fread(&eps, 4, 1, f) != 1) {
        fprintf(stderr, "bad header\n");
        return 2;
    }
    const int L = hdr[1], d = hdr[3], V = hdr[4], NP = hdr[5];
    capdec_gpt2_weights w;
    w.n_layer = L; w.n_head = hdr[2]; w.n_embd = d; w.vocab = V; w.n_pos = NP; w.ln_eps = eps;
    w.wte = read_floats(f, (size_t)V * d);
    w.wpe = read_floats(f, (size_t)NP * d);
    capdec_gpt2_layer *ly = (capdec_gpt2_layer *)calloc((size_t)L, sizeof(*ly));
    for (int i = 0; i < L; ++i) {
        ly[i].ln_1_w = read_floats(f, d);            ly[i].ln_1_b = read_floats(f, d);
        ly[i].c_attn_w = read_floats(f, (size_t)d * 3 * d);  ly[i].c_attn_b = read_floats(f, 3 * (size_t)d);
        ly[i].c_proj_w = read_floats(f, (size_t)d * d);      ly[i].c_proj_b = read_floats(f, d);
        ly[i].ln_2_w = read_floats(f, d);            ly[i].ln_2_b = read_floats(f, d);
        ly[i].c_fc_w = read_floats(f, (size_t)d * 4 * d);    ly[i].c_fc_b = read_floats(f, 4 * (size_t)d);
        ly[i].mlp_c_proj_w = read_floats(f, (size_t)4 * d * d);  ly[i].mlp_c_proj_b = read_floats(f, d);
    }
    w.layers = ly;
    w.ln_f_w = read_floats(f, d);
    w.ln_f_b = read_floats(f, d);
    int32_t np[2];
    if (fread(np, 4, 2, f) != 2) { fprintf(stderr, "no prefix block\n"); return 2; }
    const int n_total = np[0], P = np[1];
    float *prefix_all = read_floats(f, (size_t)n_total * P * d);
    fclose(f);

    capdec_ctx *ctx = NULL;
    CHECK(capdec_create(idfile ? rank : 0, &ctx));
    CHECK(capdec_load_gpt2(ctx, &w));
    int lo = 0, hi = n_total;
    if (idfile) {   /* one RCCL communicator over the ranks; the 128-byte id travels through a file */
        char id[CAPDEC_COMM_ID_BYTES];
        if (rank == 0) {
            char tmp[4096];
            CHECK(capdec_comm_unique_id(id));
            snprintf(tmp, sizeof tmp, "%s.tmp", idfile);
            FILE *g = fopen(tmp, "wb");
            if (!g || fwrite(id, 1, sizeof id, g) != sizeof id) { perror("idfile"); return 2; }
            fclose(g);
            rename(tmp, idfile);
        } else {
            FILE *g = NULL;
            for (int tries = 0; tries < 600 && !(g = fopen(idfile, "rb")); ++tries) usleep(100000);
            if (!g || fread(id, 1, sizeof id, g) != sizeof id) { fprintf(stderr, "rank %d: no id file\n", rank); return 2; }
            fclose(g);
        }
        CHECK(capdec_comm_init(ctx, rank, nranks, id));
        CHECK(capdec_shard_bounds(n_total, rank, nranks, &lo, &hi));
    }
    const int n = hi - lo;
    const float *prefix = prefix_all + (size_t)lo * P * d;
    void *d_prefix, *d_ids, *d_lens, *d_bids, *d_blens, *d_scores, *g_ids = NULL, *g_lens = NULL;
    CHECK(capdec_malloc(ctx, (size_t)n * P * d * 4, &d_prefix));
    CHECK(capdec_malloc(ctx, (size_t)n * T * 4, &d_ids));
    CHECK(capdec_malloc(ctx, (size_t)n * 4, &d_lens));
    CHECK(capdec_malloc(ctx, (size_t)n * beam * T * 4, &d_bids));
    CHECK(capdec_malloc(ctx, (size_t)n * beam * 4, &d_blens));
    CHECK(capdec_malloc(ctx, (size_t)n * beam * 4, &d_scores));
    CHECK(capdec_memcpy_h2d(ctx, d_prefix, prefix, (size_t)n * P * d * 4));
    /* generate2 (reference gpt2_prefix_eval.py:118-198): stop on '.' (13) or 764 */
    CHECK(capdec_decode_greedy(ctx, (const float *)d_prefix, n, P, 13, 764, T, (int32_t *)d_ids, (int32_t *)d_lens));
    /* generate_beam (:50-115) */
    CHECK(capdec_decode_beam(ctx, (const float *)d_prefix, n, P, beam, 13, T, 1.0f, (int32_t *)d_bids, (int32_t *)d_blens,
                             (float *)d_scores, NULL));
    if (idfile) {   /* the one exchange of the path: greedy ids / lengths of every rank, in caption order */
        CHECK(capdec_malloc(ctx, (size_t)n_total * T * 4, &g_ids));
        CHECK(capdec_malloc(ctx, (size_t)n_total * 4, &g_lens));
        CHECK(capdec_gather_ids(ctx, (const int32_t *)d_ids, (const int32_t *)d_lens, NULL, n, T, n_total,
                                (int32_t *)g_ids, (int32_t *)g_lens, NULL));
        if (rank == 0) {
            int32_t *gi = (int32_t *)malloc((size_t)n_total * T * 4), *gl = (int32_t *)malloc((size_t)n_total * 4);
            CHECK(capdec_memcpy_d2h(ctx, gi, g_ids, (size_t)n_total * T * 4));
            CHECK(capdec_memcpy_d2h(ctx, gl, g_lens, (size_t)n_total * 4));
            for (int r = 0; r < n_total; ++r) {
                printf("gathered %d :", gl[r]);
                for (int t = 0; t < gl[r]; ++t) printf(" %d", gi[(size_t)r * T + t]);
                printf("\n");
            }
            free(gi); free(gl);
        }
        CHECK(capdec_comm_destroy(ctx));
    }
    int32_t *ids = (int32_t *)malloc((size_t)n * T * 4), *lens = (int32_t *)malloc((size_t)n * 4);
    int32_t *bids = (int32_t *)malloc((size_t)n * beam * T * 4), *blens = (int32_t *)malloc((size_t)n * beam * 4);
    float *scores = (float *)malloc((size_t)n * beam * 4);
    CHECK(capdec_memcpy_d2h(ctx, ids, d_ids, (size_t)n * T * 4));
    CHECK(capdec_memcpy_d2h(ctx, lens, d_lens, (size_t)n * 4));
    CHECK(capdec_memcpy_d2h(ctx, bids, d_bids, (size_t)n * beam * T * 4));
    CHECK(capdec_memcpy_d2h(ctx, blens, d_blens, (size_t)n * beam * 4));
    CHECK(capdec_memcpy_d2h(ctx, scores, d_scores, (size_t)n * beam * 4));
    for (int r = 0; r < n; ++r) {
        printf("greedy %d :", lens[r]);
        for (int t = 0; t < lens[r]; ++t) printf(" %d", ids[(size_t)r * T + t]);
        printf("\nbeam %d %.6f :", blens[(size_t)r * beam], scores[(size_t)r * beam]);
        for (int t = 0; t < blens[(size_t)r * beam]; ++t) printf(" %d", bids[(size_t)r * beam * T + t]);
        printf("\n");
    }
    CHECK(capdec_free(ctx, d_prefix)); CHECK(capdec_free(ctx, d_ids)); CHECK(capdec_free(ctx, d_lens));
    CHECK(capdec_free(ctx, d_bids)); CHECK(capdec_free(ctx, d_blens)); CHECK(capdec_free(ctx, d_scores));
    CHECK(capdec_free(ctx, g_ids)); CHECK(capdec_free(ctx, g_lens));
    capdec_destroy(ctx);
    return 0;
}
