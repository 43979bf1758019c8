/* gbp_mi355x_multi.h — the landmark-sharded (multi-GPU) half of the C-ABI of the MI355X-native GBP engine.
 *
 * Replaces `--ipus N` of the reference (ba/ba.cpp:414-417,617-649: one Poplar graph over N IPUs, the inter-IPU exchange compiled
 * into the program).  Here: one process per GPU, each with a SHARDED ctx (gbp_create with a gbp_shard: a contiguous landmark
 * range and every factor incident to it; cameras replicated), ONE all-gather of the [C x 44] camera partial sums per iteration.
 * Two ways to run the exchange:
 *   - owned by the library (gbp_comm_*): RCCL over xGMI, dlopen'ed, or a host-staged transport for ranks that share a GPU; once a
 *     ctx has a communicator the plain program list of gbp_mi355x.h works on it;
 *   - owned by the caller (split-phase: gbp_iterate_begin / exchange / gbp_iterate_end), e.g. torch.distributed.
 * A single-GPU host needs nothing from this header (INTEGRATION.md binds the core header only).
 */
#ifndef GBP_MI355X_MULTI_H
#define GBP_MI355X_MULTI_H

#include "gbp_mi355x.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- multi-GPU split-phase iteration (sharded ctx; exchange done by the caller, e.g.
 *      torch.distributed all_gather over RCCL).  gbp_iterate == begin + (local copy) + end
 *      when world == 1. -------------------------------------------------------------------- */
GBP_API int gbp_set_stream(gbp_ctx* ctx, void* hip_stream /* hipStream_t; NULL = ctx-owned stream */);
/* send_dev: [C*GBP_CAM_REC] fp32 this rank's camera partial sums; recv_dev: [world][C][GBP_CAM_REC]
 * (camera record = 44 floats: eta 6, pad 2, Lambda 36).  Caller-owned device memory (e.g. torch
 * tensors).  Must be set before begin/end on a world>1 ctx. */
#define GBP_CAM_REC 44
GBP_API int gbp_set_exchange_buffers(gbp_ctx* ctx, void* send_dev, void* recv_dev);
GBP_API int gbp_iterate_begin(gbp_ctx* ctx);   /* prep + messages + local camera partial -> send_dev     */
GBP_API int gbp_iterate_local(gbp_ctx* ctx);   /* optional: landmark beliefs now (rank-local), to overlap with the exchange */
GBP_API int gbp_iterate_end(gbp_ctx* ctx);     /* camera beliefs = prior + sum_r recv_dev[r]; landmark beliefs unless done */
/* Re-derive beliefs after an exchange outside an iteration (LINEARISE / NEW_KEYFRAME on world>1):
 * gbp_refresh_begin computes the local camera partials into send_dev, gbp_refresh_end combines. */
GBP_API int gbp_refresh_begin(gbp_ctx* ctx);
GBP_API int gbp_refresh_end(gbp_ctx* ctx);
GBP_API int gbp_linearise_factors(gbp_ctx* ctx);  /* the factor half of LINEARISE_PROG (after a refresh) */

/* ---- multi-GPU from the C++ host: the exchange owned by the library ---------------------------------------------------
 * Replaces `--ipus N` (ba.cpp:414-417,617-649) without any Python: one process per GPU, each with a sharded ctx
 * (gbp_shard).  Once a ctx has a communicator, the plain program list works on it — gbp_linearise, gbp_iterate(n)
 * (sweep -> local camera partials -> ncclAllGather on a second stream, overlapped with the rank-local landmark beliefs
 * -> camera combine; captured into a hipGraph like the single-GPU iteration), gbp_weaken_priors, gbp_new_keyframe —
 * and gbp_eval_global adds the metric sums of all shards in rank order.
 *
 * Launchers that fork their ranks (bin/ba --ipus N) share one MAP_SHARED region, created and initialised BEFORE the
 * ranks start; it carries the rendezvous (RCCL unique id, barrier) and, for ranks that share a GPU (fewer GPUs than
 * ranks — RCCL refuses duplicate GPUs), the host-staged transport that moves the same buffers through host memory.
 * transport: 0 = auto (RCCL when every rank has its own GPU, host-staged otherwise), 1 = RCCL, 2 = host-staged.
 * Launchers with their own rendezvous (torchrun, MPI) pass the 128-byte RCCL id around themselves:
 * gbp_comm_unique_id on rank 0, gbp_comm_init_rccl on every rank.  All calls are collective over the ranks.
 * Scheduling: with 4 ranks or more the camera side of the exchange (local partial sums, all-gather) runs on a second,
 * highest-priority stream beside the landmark beliefs; with 1 or 2 ranks everything stays on one stream (a second HSA
 * queue costs more per dispatch than a small all-gather gives back).  Environment GBP_COMM_SINGLE_STREAM=0|1, read by
 * gbp_comm_init*, overrides the rule (measurements, tests). */
#define GBP_COMM_ID_BYTES 128
GBP_API int gbp_device_count(void);                                    /* visible GPUs (initialises the HIP runtime)      */
GBP_API int gbp_set_device(int device);                                /* the GPU later gbp_create calls of this process use */
/* contiguous landmark ranges balanced by factor count: bounds[world + 1], shard r = [bounds[r], bounds[r+1]) */
GBP_API int gbp_landmark_partition(const gbp_problem* problem, int world, uint32_t* bounds);
GBP_API size_t gbp_comm_region_bytes(uint32_t n_cams, int world);
GBP_API int gbp_comm_region_init(void* region, size_t bytes, uint32_t n_cams, int world);
GBP_API void gbp_comm_region_abort(void* region);                      /* supervisor: a rank died, fail the waiting ones */
/* the region's cross-process protocol alone (gathers + barriers, no device): every rank of `world` calls it; test hook */
GBP_API int gbp_comm_region_selftest(void* region, int rank, int world, int rounds);
GBP_API int gbp_comm_init(gbp_ctx* ctx, void* region, int transport);
GBP_API int gbp_comm_unique_id(void* id128);
GBP_API int gbp_comm_init_rccl(gbp_ctx* ctx, const void* id128);
GBP_API const char* gbp_comm_transport(const gbp_ctx* ctx);            /* "rccl", "host-staged" or "none" */
GBP_API int gbp_comm_barrier(gbp_ctx* ctx);
/* What a first multi-GPU run puts on record next to its numbers (bench.py's preflight block): gbp_comm_describe writes one
 * JSON object (rank, world, device, PCI bus id, transport, the collective library's resolved path and version, schedule);
 * gbp_comm_probe times `reps` all-gathers of the camera partial buffers back to back (collective); gbp_comm_set_schedule
 * switches between the one-stream and the two-stream form of the sharded iteration (identical results) so that a launcher can
 * MEASURE both and keep the faster one instead of trusting the ">= 4 ranks" rule (ba.cpp:617-649 has no such choice to make:
 * Poplar compiles the exchange into the program). */
GBP_API int gbp_comm_describe(gbp_ctx* ctx, char* json_buf, size_t cap);
GBP_API int gbp_comm_set_schedule(gbp_ctx* ctx, int two_streams);
GBP_API int gbp_comm_probe(gbp_ctx* ctx, int reps, double* avg_us);
GBP_API int gbp_eval_global(gbp_ctx* ctx, gbp_eval_out* out);          /* gbp_eval summed over all shards */

#ifdef __cplusplus
}
#endif
#endif
