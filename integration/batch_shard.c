/* The C shard host of BASELINE.json's north star: "many independent IQ captures are batched as the data-parallel axis and sharded
 * across the GPUs of one node ... host stays in C".  Plain C over include/nrsc5hip.h, no HIP headers, no Python:
 *
 *     batch_shard <captures.cu8> <nstreams> <bytes_per_stream> [gpus]
 *
 * The file holds `nstreams` cu8 captures of `bytes_per_stream` bytes back to back.  One host thread and ONE ENGINE per visible GPU
 * (or the first `gpus`); stream k goes to GPU k mod N -- an embarrassingly parallel split, no collective on the data path -- through
 * the batch entry points (captures uploaded once, read in place by the block steps, window pipeline, on-device L2 feedback: the
 * configuration bench.py measures).  Prints one line per stream, in stream order:
 *
 *     stream <k> gpu <g> blocks <B> p1 <F> pids <P> fine <N> crc32 <xxxxxxxx>
 *
 * where crc32 runs over the stream's packed P1 frames in record order -- the same per-stream summary bench.py's ranks gather
 * (Fm.verify), so tests/test_gpu_dropin.py::test_batch_shard_c_host_equals_python_path compares the two byte for byte.
 * Replaces nothing of the reference: its host side has no batch mode (src/main.c feeds one capture).  Build: integration/Makefile. */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "nrsc5hip.h"

#define BLOCK_BYTES   276480u            /* cu8 bytes per 32-symbol FM block: 32 * 2160 * 2 samples * 2 bytes */
#define FRAME_BYTES   (16u * BLOCK_BYTES)

typedef struct {
    int gpu, ngpu, nstreams_total;
    const uint8_t *file; size_t bytes_per_stream;
    /* results, indexed by global stream */
    int *blocks, *p1, *pids, *fine; uint32_t *crc;
    int rc; char err[512];
} shard_t;

static uint32_t crc_table[256];
static void crc_init(void)
{
    for (uint32_t i = 0; i < 256; i++) { uint32_t c = i; for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1; crc_table[i] = c; }
}
static uint32_t crc32_update(uint32_t crc, const void *buf, size_t n)       /* zlib.crc32(buf, crc) */
{
    const uint8_t *p = (const uint8_t *)buf;
    crc = ~crc;
    while (n--) crc = crc_table[(crc ^ *p++) & 0xff] ^ (crc >> 8);
    return ~crc;
}

#define CHECK(call) do { int _rc = (call); if (_rc) { sh->rc = _rc; snprintf(sh->err, sizeof(sh->err), "%s -> %d: %s", #call, _rc, nrsc5hip_last_error()); goto out; } } while (0)

static void *shard_main(void *arg)
{
    shard_t *sh = (shard_t *)arg;
    nrsc5hip_engine *e = NULL;
    void *dev = NULL;
    uint32_t *nbytes = NULL;
    /* my streams: k = gpu, gpu + N, gpu + 2 N, ... */
    int mine = 0;
    for (int k = sh->gpu; k < sh->nstreams_total; k += sh->ngpu) mine++;
    if (!mine) return NULL;
    const size_t stride = (sh->bytes_per_stream + 255) & ~(size_t)255;
    const int nframes = (int)(sh->bytes_per_stream / FRAME_BYTES) + 1;
    nrsc5hip_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.device = sh->gpu; cfg.max_streams = mine; cfg.q15_capacity = 2 * 71280;     /* zero-copy: the FIFO stays at its minimum */
    cfg.record_capacity = 2 * 16 * nframes + 64 < 512 ? 512 : 2 * 16 * nframes + 64;
    cfg.p1_slots = nframes + 12; cfg.p1_async = 1; cfg.l2_feedback = 1; cfg.batch_zero_copy = 1;
    CHECK(nrsc5hip_engine_create(&cfg, &e));
    {
        /* captures of my streams in one strided host image -> one upload (local stream j = global stream gpu + j N) */
        uint8_t *img = (uint8_t *)calloc((size_t)mine, stride);
        if (!img) { sh->rc = NRSC5HIP_ENOMEM; snprintf(sh->err, sizeof(sh->err), "out of host memory"); goto out; }
        for (int j = 0; j < mine; j++) memcpy(img + (size_t)j * stride, sh->file + (size_t)(sh->gpu + j * sh->ngpu) * sh->bytes_per_stream, sh->bytes_per_stream);
        int rc = nrsc5hip_device_upload(sh->gpu, img, stride * (size_t)mine, &dev);
        free(img);
        if (rc) { sh->rc = rc; snprintf(sh->err, sizeof(sh->err), "upload: %s", nrsc5hip_last_error()); goto out; }
    }
    nbytes = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)mine);
    for (int j = 0; j < mine; j++) nbytes[j] = (uint32_t)(sh->bytes_per_stream & ~(size_t)3);
    CHECK(nrsc5hip_batch_append_cu8(e, mine, NULL, (const uint8_t *)dev, (long long)stride, nbytes));
    int steps = 0;
    CHECK(nrsc5hip_batch_process(e, mine, NULL, 0, &steps));
    {
        const nrsc5hip_record *recs = NULL; const uint32_t *frames = NULL;
        int *counts = (int *)calloc((size_t)mine, sizeof(int));
        int rc = nrsc5hip_batch_fetch_view(e, mine, &recs, counts, &frames);
        if (rc) { free(counts); sh->rc = rc; snprintf(sh->err, sizeof(sh->err), "fetch: %s", nrsc5hip_last_error()); goto out; }
        for (int j = 0; j < mine; j++) {
            const int k = sh->gpu + j * sh->ngpu;
            const nrsc5hip_record *r = recs + (size_t)j * cfg.record_capacity;
            uint32_t crc = 0; int p1 = 0, pids = 0, fine = 0;
            for (int i = 0; i < counts[j]; i++) {
                if (r[i].flags & NRSC5HIP_REC_PIDS) pids++;
                if (r[i].state_after == NRSC5HIP_SYNC_FINE) fine++;
                if (r[i].flags & NRSC5HIP_REC_P1) {
                    p1++;
                    crc = crc32_update(crc, frames + ((size_t)j * cfg.p1_slots + (size_t)r[i].p1_slot) * NRSC5HIP_P1_FRAME_WORDS, NRSC5HIP_P1_FRAME_WORDS * 4);
                }
            }
            sh->blocks[k] = counts[j]; sh->p1[k] = p1; sh->pids[k] = pids; sh->fine[k] = fine; sh->crc[k] = crc;
        }
        free(counts);
    }
out:
    free(nbytes);
    if (e) nrsc5hip_engine_destroy(e);
    if (dev) (void)nrsc5hip_device_free(sh->gpu, dev);
    return NULL;
}

int main(int argc, char **argv)
{
    if (argc < 4) { fprintf(stderr, "usage: %s <captures.cu8> <nstreams> <bytes_per_stream> [gpus]\n", argv[0]); return 2; }
    const int nstreams = atoi(argv[2]);
    const size_t per = (size_t)strtoull(argv[3], NULL, 10);
    int ngpu = 0;
    if (nrsc5hip_device_count(&ngpu) || ngpu < 1) { fprintf(stderr, "no GPU: %s\n", nrsc5hip_last_error()); return 1; }
    if (argc > 4 && atoi(argv[4]) > 0 && atoi(argv[4]) < ngpu) ngpu = atoi(argv[4]);
    if (nstreams < 1 || per < 4) { fprintf(stderr, "bad stream count / size\n"); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    uint8_t *file = (uint8_t *)malloc(per * (size_t)nstreams);
    if (!file || fread(file, 1, per * (size_t)nstreams, f) != per * (size_t)nstreams) { fprintf(stderr, "short read\n"); return 1; }
    fclose(f);
    crc_init();
    int *blocks = calloc((size_t)nstreams, sizeof(int)), *p1 = calloc((size_t)nstreams, sizeof(int)), *pids = calloc((size_t)nstreams, sizeof(int)), *fine = calloc((size_t)nstreams, sizeof(int));
    uint32_t *crc = calloc((size_t)nstreams, sizeof(uint32_t));
    shard_t *sh = calloc((size_t)ngpu, sizeof(shard_t));
    pthread_t *th = calloc((size_t)ngpu, sizeof(pthread_t));
    for (int g = 0; g < ngpu; g++) {
        sh[g] = (shard_t){ .gpu = g, .ngpu = ngpu, .nstreams_total = nstreams, .file = file, .bytes_per_stream = per,
                           .blocks = blocks, .p1 = p1, .pids = pids, .fine = fine, .crc = crc };
        pthread_create(&th[g], NULL, shard_main, &sh[g]);
    }
    int bad = 0;
    for (int g = 0; g < ngpu; g++) { pthread_join(th[g], NULL); if (sh[g].rc) { fprintf(stderr, "gpu %d: %s\n", g, sh[g].err); bad = 1; } }
    if (bad) return 1;
    for (int k = 0; k < nstreams; k++)
        printf("stream %d gpu %d blocks %d p1 %d pids %d fine %d crc32 %08x\n", k, k % ngpu, blocks[k], p1[k], pids[k], fine[k], crc[k]);
    return 0;
}
