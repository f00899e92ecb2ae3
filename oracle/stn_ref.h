/* ORACLE header — test infrastructure (see stn_ref.c).  Plain C ABI for ctypes. */
#ifndef STN_REF_H
#define STN_REF_H
#include <stdint.h>
#include "../include/stn_arch.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct stnref_model stnref_model;

void stnref_set_threads(int n);
void stnref_set_gelu_tanh(int on);  /* process-wide: 0 = erf form (default), 1 = the tanh approximation */
int stnref_get_threads(void);
stnref_model* stnref_create(const stn_arch* a, uint64_t seed);
void stnref_destroy(stnref_model* m);
int64_t stnref_param_count(const stnref_model* m);
int stnref_num_tensors(const stnref_model* m);
const char* stnref_tensor_name(const stnref_model* m, int i);
int64_t stnref_tensor(const stnref_model* m, const char* name, float* out, int64_t cap);

/* the four stages; argument meaning == the four Run sites of cpp/helper.cpp */
void stnref_duration(const stnref_model* m, int B, int Lt, const int64_t* text_ids, const float* style_dp,
                     const float* text_mask, float* duration);
void stnref_text_enc(const stnref_model* m, int B, int Lt, const int64_t* text_ids, const float* style_ttl,
                     const float* text_mask, float* text_emb);
void stnref_vector_est(const stnref_model* m, int B, int L, int Lt, const float* noisy_latent, const float* text_emb,
                       const float* style_ttl, const float* text_mask, const float* latent_mask,
                       const float* total_step, const float* current_step, float* denoised);
void stnref_vocoder(const stnref_model* m, int B, int L, const float* latent, float* wav);

/* op level (kernel unit tests) */
void stnref_linear(const float* X, int64_t M, int K, const float* Wt, const float* bias, int N, float* Y);
void stnref_layernorm(const float* X, int64_t M, int C, const float* g, const float* b, float eps, float* Y);
void stnref_dwconv(const float* X, int B, int L, int C, const float* w, const float* bias, int k, int dil, float* Y);
void stnref_attention_core(const float* Q, const float* K, const float* V, int B, int Lq, int Lk, int C, int H,
                           const int* klen, float* O);
void stnref_randn(uint64_t seed, int B, int D, int L, const int64_t* utt_ids, float* out);

#ifdef __cplusplus
}
#endif
#endif
