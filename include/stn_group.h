/* stn_group.h — several MI355X in ONE process behind the C ABI (same library, libstn.so).
 *
 * north_star: "independent utterances in a batch shard embarrassingly across the 8 GPUs of one node (RCCL over xGMI only to gather
 * finished waveforms)".  The reference's batch is only a leading dimension (/root/reference/cpp/helper.cpp:477) and its host calls
 * _infer once per batch (cpp/example_onnx.cpp:88-97), so a C++ / cgo / JNI caller of include/stn.h gets the split with one call
 * instead of writing it:
 *   - one engine handle (stn.h), one HIP stream and one worker thread per device, weights replicated;
 *   - utterances sorted by length (descending, stable) and dealt round-robin — the rule of supertonic_amd/dist.py:shard_by_length,
 *     SURVEY.md section 8(e) — so every device gets the same share of long and short utterances; noise is keyed by the utterance's
 *     index in the CALLER's batch, so a dealt batch draws the noise of the undealt one;
 *   - every device runs DP -> text encoder -> noise -> Euler steps -> vocoder on its shard and converts to 16-bit PCM (the
 *     reference's final product: writeWavFile, cpp/helper.cpp:943-990) on the GPU;
 *   - ONE exchange: the PCM blocks travel to the first device over RCCL — ncclCommInitAll once at stn_group_create, then per
 *     synthesis ncclGroupStart; ncclSend (rank r -> 0) / ncclRecv (0 <- r), r = 1..n-1; ncclGroupEnd on the engines' own streams, so
 *     the exchange is ordered behind each device's kernels with no host synchronisation — and from there to the host in caller order.
 *     There is no data-path collective anywhere else.  librccl is loaded (dlopen) only by groups of more than one distinct device.
 * A device ordinal listed more than once is accepted as a REHEARSAL of the multi-rank path on fewer GPUs (one-GPU test boxes): the
 * ranks then share a GPU and the exchange is a device-to-device copy ordered by events; deal, threads, block layout and the reorder
 * into caller order are the ones of the real path.
 *
 * Return codes and error text as in stn.h (stn_group_last_error).  Calls on one group are serialised by the caller.
 */
#ifndef STN_GROUP_H
#define STN_GROUP_H
#include <stddef.h>
#include <stdint.h>

#include "stn.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct stn_group stn_group;

/* n_devices >= 1; devices_or_null: HIP ordinals (NULL: 0 .. n_devices-1).  More devices than the box has: STN_ERR_DEVICE and a
 * message naming both numbers (stn_group_last_error(NULL)). */
int stn_group_create(int n_devices, const int* devices_or_null, int dtype, stn_group** out);
int stn_group_destroy(stn_group* g);
const char* stn_group_last_error(const stn_group* g_or_null);
int stn_group_size(const stn_group* g);
/* 1 when the gather runs over RCCL (more than one distinct device; or one device under the measurement switch STN_GROUP_SELF_RCCL=1, which sends the
 * block to itself), 0 for one device or a rehearsal on shared devices */
int stn_group_uses_rccl(const stn_group* g);
/* rank r's engine handle (settings, diagnostics); owned by the group */
stn_handle* stn_group_handle(stn_group* g, int rank);
/* the same model on every device */
int stn_group_load_synthetic(stn_group* g, const stn_arch* arch, uint64_t seed);
int stn_group_load_dir(stn_group* g, const char* onnx_dir);

/* The deal, host only: utterance i goes to rank rank_of[i] as row row_of[i] of that rank's shard.  lengths[B] = token counts.
 * Sorted by length descending (ties: caller order), dealt round-robin: the k-th longest goes to rank k % n as row k / n. */
int stn_group_deal(int B, const int32_t* lengths, int n_ranks, int32_t* rank_of, int32_t* row_of);

/* One synthesis of B utterances over the group's devices: inputs as stn_batch_upload / stn_batch_run (host pointers, caller order).
 * Returns when every shard's PCM is in the first device's memory and the durations are known; *samples_per_utt_out = the row length
 * W of the result (the longest shard's L * chunk samples; shorter shards' rows are zero-filled behind their own W_r). */
int stn_group_synthesize(stn_group* g, int B, int Lt, const int64_t* text_ids, const float* text_mask, const float* style_ttl,
                         const float* style_dp, int total_step, float speed, const float* duration_override_or_null, uint64_t noise_seed,
                         int64_t* samples_per_utt_out);
/* the result of the last synthesis in CALLER order: pcm [B][W] int16 (capacity in samples), duration [B] seconds (after /speed) */
int stn_group_fetch_pcm16(stn_group* g, int16_t* pcm, size_t capacity_samples, float* duration);
/* how the last synthesis was dealt: utterances and samples per utterance of every rank's block (n values each) */
int stn_group_last_shards(const stn_group* g, int32_t* rows_per_rank, int64_t* samples_per_rank);

#ifdef __cplusplus
}
#endif
#endif /* STN_GROUP_H */
