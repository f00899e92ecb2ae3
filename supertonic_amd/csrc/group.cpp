// group.cpp — include/stn_group.h: n devices in one process on top of the C ABI of stn.h (one handle, stream and worker thread per
// device; length-sorted round-robin deal; 16-bit PCM gathered into the first device over RCCL; caller-order result on the host).
// Stands in for the batch dimension of /root/reference/cpp/helper.cpp:477 spread over the GPUs of a node.
#include "../../include/stn_group.h"
#include "dev_env.hpp"

#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <numeric>
#include <set>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace {

thread_local std::string g_group_create_err;

// librccl, loaded on demand: only a group of more than one distinct device needs it
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string open() {
        if (lib) return "";
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) return std::string("librccl not found (dlopen): ") + (dlerror() ? dlerror() : "");
        auto sym = [&](const char* n) { return dlsym(lib, n); };
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
        Send = reinterpret_cast<decltype(Send)>(sym("ncclSend"));
        Recv = reinterpret_cast<decltype(Recv)>(sym("ncclRecv"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
        if (!CommInitAll || !CommDestroy || !GroupStart || !GroupEnd || !Send || !Recv || !GetErrorString) return "librccl lacks a symbol this library needs";
        return "";
    }
};

struct Rank {
    int device = 0;
    stn_handle* h = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;     // rehearsal: the shard's PCM is in `send`
    void* send = nullptr;          // this rank's PCM block [B_r][W_r] int16, on its device
    size_t send_cap = 0;
    void* recv = nullptr;          // ranks >= 1: the block's landing place on the first device
    size_t recv_cap = 0;
    int B = 0;                     // last synthesis: utterances, samples per utterance
    int64_t W = 0;
    std::vector<int> idx;          // caller indices of the shard's rows
    std::vector<float> dur;
    std::string err;
    int rc = STN_OK;
};

}  // namespace

struct stn_group {
    std::vector<Rank> ranks;
    bool rccl = false;
    bool self_rccl = false;  // measurement switch STN_GROUP_SELF_RCCL=1 on a group of one: the root's own block travels through ncclSend / ncclRecv to itself,
                             // so a one-GPU box executes the library calls of the exchange (dlopen, ncclCommInitAll, grouped send / receive on the stream)
    Rccl nccl;
    std::vector<ncclComm_t> comms;
    void* host_stage = nullptr;  // pinned: all blocks back to back
    size_t host_cap = 0;
    int B = 0;
    int64_t W = 0;
    std::string err;
};

namespace {

#define HIPG(call)                                                                                              \
    do {                                                                                                        \
        const hipError_t e_ = (call);                                                                           \
        if (e_ != hipSuccess) { (void)hipGetLastError(); throw std::runtime_error(std::string("HIP error: ") + #call + ": " + hipGetErrorString(e_)); } \
    } while (0)

void grow(void** p, size_t* cap, size_t need) {
    if (need <= *cap) return;
    if (*p) HIPG(hipFree(*p));
    *p = nullptr;
    *cap = 0;
    const size_t n = need + need / 4;
    HIPG(hipMalloc(p, n));
    *cap = n;
}

void deal(int B, const int32_t* lengths, int n, int32_t* rank_of, int32_t* row_of) {
    std::vector<int> order(B);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return lengths[a] > lengths[b]; });
    for (int k = 0; k < B; ++k) { rank_of[order[k]] = k % n; row_of[order[k]] = k / n; }
}

int fail(stn_group* g, int code, const std::string& msg) {
    if (g) g->err = msg; else g_group_create_err = msg;
    return code;
}

}  // namespace

extern "C" {

const char* stn_group_last_error(const stn_group* g) { return g ? g->err.c_str() : g_group_create_err.c_str(); }
int stn_group_size(const stn_group* g) { return g ? (int)g->ranks.size() : STN_ERR_INVALID; }
int stn_group_uses_rccl(const stn_group* g) { return g ? (g->rccl ? 1 : 0) : STN_ERR_INVALID; }
stn_handle* stn_group_handle(stn_group* g, int rank) { return (g && rank >= 0 && rank < (int)g->ranks.size()) ? g->ranks[rank].h : nullptr; }

int stn_group_deal(int B, const int32_t* lengths, int n_ranks, int32_t* rank_of, int32_t* row_of) {
    if (B < 0 || n_ranks < 1 || (B > 0 && (!lengths || !rank_of || !row_of))) return STN_ERR_INVALID;
    deal(B, lengths, n_ranks, rank_of, row_of);
    return STN_OK;
}

int stn_group_destroy(stn_group* g) {
    if (!g) return STN_ERR_INVALID;
    for (size_t r = 0; r < g->ranks.size(); ++r) {
        Rank& k = g->ranks[r];
        if (k.h) (void)stn_sync(k.h);
    }
    if (g->rccl) for (ncclComm_t c : g->comms) if (c) (void)g->nccl.CommDestroy(c);
    for (Rank& k : g->ranks) {
        (void)hipSetDevice(k.device);
        if (k.h) (void)stn_destroy(k.h);  // (the handle leaves the caller-owned stream alone)
        if (k.send) (void)hipFree(k.send);
        if (k.done) (void)hipEventDestroy(k.done);
        if (k.stream) (void)hipStreamDestroy(k.stream);
    }
    if (!g->ranks.empty()) {
        (void)hipSetDevice(g->ranks[0].device);
        for (Rank& k : g->ranks) if (k.recv) (void)hipFree(k.recv);
    }
    if (g->host_stage) (void)hipHostFree(g->host_stage);
    delete g;
    return STN_OK;
}

int stn_group_create(int n_devices, const int* devices, int dtype, stn_group** out) {
    if (!out || n_devices < 1 || n_devices > 64) return fail(nullptr, STN_ERR_INVALID, "stn_group_create: n_devices must be 1..64 and out non-null");
    *out = nullptr;
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess) { (void)hipGetLastError(); have = 0; }
    std::vector<int> dev(n_devices);
    for (int i = 0; i < n_devices; ++i) dev[i] = devices ? devices[i] : i;
    const std::set<int> distinct(dev.begin(), dev.end());
    for (int d : dev)
        if (d < 0 || d >= have)
            return fail(nullptr, STN_ERR_DEVICE, "stn_group_create: the group needs HIP device " + std::to_string(d) + " (" + std::to_string((int)distinct.size()) +
                                                     " distinct device(s) asked for) but only " + std::to_string(have) + " are visible");
    stn_group* g = new stn_group;
    g->ranks.resize(n_devices);
    try {
        for (int r = 0; r < n_devices; ++r) {
            Rank& k = g->ranks[r];
            k.device = dev[r];
            HIPG(hipSetDevice(k.device));
            HIPG(hipStreamCreateWithFlags(&k.stream, hipStreamNonBlocking));
            HIPG(hipEventCreateWithFlags(&k.done, hipEventDisableTiming));
            stn_config cfg{k.device, dtype};
            const int rc = stn_create(&cfg, &k.h);
            if (rc != STN_OK) { const std::string m = std::string("stn_create on device ") + std::to_string(k.device) + ": " + stn_last_error(nullptr); stn_group_destroy(g); return fail(nullptr, rc, m); }
            if (stn_set_stream(k.h, k.stream) != STN_OK) throw std::runtime_error(stn_last_error(k.h));
        }
        g->rccl = n_devices > 1 && (int)distinct.size() == n_devices;
        if (n_devices == 1) { const char* e = stn::dev_env("STN_GROUP_SELF_RCCL"); g->self_rccl = g->rccl = e && atoi(e) == 1; }
        if (n_devices > 1 && !g->rccl && distinct.size() != 1) {
            // (a mix of shared and distinct devices would need both exchange forms at once: not a configuration anyone runs)
            stn_group_destroy(g);
            return fail(nullptr, STN_ERR_INVALID, "stn_group_create: device ordinals must be all distinct (RCCL gather) or all the same (rehearsal on one GPU)");
        }
        if (g->rccl) {
            const std::string e = g->nccl.open();
            if (!e.empty()) { stn_group_destroy(g); return fail(nullptr, STN_ERR_DEVICE, "stn_group_create: " + e); }
            g->comms.assign(n_devices, nullptr);
            const ncclResult_t nr = g->nccl.CommInitAll(g->comms.data(), n_devices, dev.data());
            if (nr != ncclSuccess) { const std::string m = std::string("ncclCommInitAll: ") + g->nccl.GetErrorString(nr); g->comms.clear(); g->rccl = false; stn_group_destroy(g); return fail(nullptr, STN_ERR_DEVICE, m); }
        }
    } catch (const std::exception& e) {
        const std::string m = e.what();
        stn_group_destroy(g);
        return fail(nullptr, STN_ERR_DEVICE, "stn_group_create: " + m);
    }
    *out = g;
    return STN_OK;
}

static int for_all(stn_group* g, const char* what, int (*fn)(stn_handle*, const void*, uint64_t), const void* a, uint64_t b) {
    if (!g) return STN_ERR_INVALID;
    std::vector<std::thread> th;
    for (Rank& k : g->ranks) th.emplace_back([&k, fn, a, b] { k.rc = fn(k.h, a, b); if (k.rc != STN_OK) k.err = stn_last_error(k.h); });
    for (auto& t : th) t.join();
    for (size_t r = 0; r < g->ranks.size(); ++r)
        if (g->ranks[r].rc != STN_OK) return fail(g, g->ranks[r].rc, std::string(what) + " on rank " + std::to_string(r) + ": " + g->ranks[r].err);
    return STN_OK;
}
int stn_group_load_synthetic(stn_group* g, const stn_arch* arch, uint64_t seed) {
    return for_all(g, "stn_load_synthetic", [](stn_handle* h, const void* a, uint64_t s) { return stn_load_synthetic(h, static_cast<const stn_arch*>(a), s); }, arch, seed);
}
int stn_group_load_dir(stn_group* g, const char* onnx_dir) {
    return for_all(g, "stn_load_dir", [](stn_handle* h, const void* a, uint64_t) { return stn_load_dir(h, static_cast<const char*>(a)); }, onnx_dir, 0);
}

int stn_group_synthesize(stn_group* g, int B, int Lt, const int64_t* text_ids, const float* text_mask, const float* style_ttl,
                         const float* style_dp, int total_step, float speed, const float* duration_override, uint64_t noise_seed,
                         int64_t* samples_per_utt_out) {
    if (!g) return STN_ERR_INVALID;
    if (B < 1 || Lt < 1 || !text_ids || !text_mask || !style_ttl || !style_dp || total_step < 1 || !(speed > 0.f))
        return fail(g, STN_ERR_INVALID, "stn_group_synthesize: B, Lt, total_step >= 1, speed > 0 and non-null inputs");
    const int n = (int)g->ranks.size();
    g->B = 0;  // (a failed synthesis leaves nothing to fetch)
    stn_arch a;
    if (stn_get_arch(g->ranks[0].h, &a) != STN_OK) return fail(g, STN_ERR_STATE, std::string("stn_group_synthesize: ") + stn_last_error(g->ranks[0].h));
    const size_t ttl_n = (size_t)a.n_style_ttl * a.d_style_ttl, dp_n = (size_t)a.n_style_dp * a.d_style_dp;
    // token counts (the mask is a prefix mask) and the deal
    std::vector<int32_t> len(B), rank_of(B), row_of(B);
    for (int i = 0; i < B; ++i) {
        int c = 0;
        for (int t = 0; t < Lt; ++t) c += text_mask[(size_t)i * Lt + t] > 0.5f;
        len[i] = c;
    }
    deal(B, len.data(), n, rank_of.data(), row_of.data());
    for (Rank& k : g->ranks) { k.idx.clear(); k.B = 0; k.W = 0; k.rc = STN_OK; k.err.clear(); }
    for (int i = 0; i < B; ++i) {
        Rank& k = g->ranks[rank_of[i]];
        if ((int)k.idx.size() <= row_of[i]) k.idx.resize(row_of[i] + 1);
        k.idx[row_of[i]] = i;
    }
    // every rank: its shard's inputs (token rows cut to the shard's longest), upload, run, PCM into its send block
    std::vector<std::thread> th;
    for (int r = 0; r < n; ++r) {
        th.emplace_back([&, r] {
            Rank& k = g->ranks[r];
            try {
                const int Br = (int)k.idx.size();
                if (Br == 0) return;  // fewer utterances than ranks
                int Ltr = 1;
                for (int i : k.idx) Ltr = std::max(Ltr, (int)len[i]);
                std::vector<int64_t> ids((size_t)Br * Ltr), utt(Br);
                std::vector<float> mask((size_t)Br * Ltr), ttl((size_t)Br * ttl_n), dp((size_t)Br * dp_n), dov(duration_override ? Br : 0);
                for (int j = 0; j < Br; ++j) {
                    const int i = k.idx[j];
                    std::memcpy(&ids[(size_t)j * Ltr], text_ids + (size_t)i * Lt, sizeof(int64_t) * Ltr);
                    std::memcpy(&mask[(size_t)j * Ltr], text_mask + (size_t)i * Lt, sizeof(float) * Ltr);
                    std::memcpy(&ttl[(size_t)j * ttl_n], style_ttl + (size_t)i * ttl_n, sizeof(float) * ttl_n);
                    std::memcpy(&dp[(size_t)j * dp_n], style_dp + (size_t)i * dp_n, sizeof(float) * dp_n);
                    utt[j] = i;  // the noise generator is keyed by the caller's index: a dealt batch draws the undealt batch's noise
                    if (duration_override) dov[j] = duration_override[i];
                }
                HIPG(hipSetDevice(k.device));
                k.rc = stn_batch_upload(k.h, Br, Ltr, ids.data(), mask.data(), ttl.data(), dp.data(), duration_override ? dov.data() : nullptr, utt.data());
                if (k.rc == STN_OK) k.rc = stn_batch_run(k.h, total_step, speed, noise_seed);
                int Bd = 0, Ld = 0;
                int64_t Wd = 0;
                if (k.rc == STN_OK) k.rc = stn_batch_dims(k.h, &Bd, &Ld, &Wd);
                if (k.rc != STN_OK) { k.err = stn_last_error(k.h); return; }
                k.B = Bd; k.W = Wd;
                grow(&k.send, &k.send_cap, (size_t)Bd * Wd * sizeof(int16_t));
                k.rc = stn_batch_copy_pcm16_device(k.h, k.send, Wd);  // enqueued on the rank's stream, behind the vocoder
                if (k.rc != STN_OK) { k.err = stn_last_error(k.h); return; }
                HIPG(hipEventRecord(k.done, k.stream));
            } catch (const std::exception& e) { k.rc = STN_ERR_DEVICE; k.err = e.what(); }
        });
    }
    for (auto& t : th) t.join();
    for (int r = 0; r < n; ++r)
        if (g->ranks[r].rc != STN_OK) return fail(g, g->ranks[r].rc, "stn_group_synthesize, rank " + std::to_string(r) + " (device " + std::to_string(g->ranks[r].device) + "): " + g->ranks[r].err);
    try {
        // ---- the one exchange: every other rank's block into the first device's memory -------------------------------------------
        Rank& root = g->ranks[0];
        HIPG(hipSetDevice(root.device));
        const int first_peer = g->self_rccl ? 0 : 1;
        for (int r = first_peer; r < n; ++r) grow(&g->ranks[r].recv, &g->ranks[r].recv_cap, (size_t)g->ranks[r].B * g->ranks[r].W * sizeof(int16_t));
        if (g->rccl) {
            ncclResult_t nr = g->nccl.GroupStart();
            for (int r = first_peer; r < n && nr == ncclSuccess; ++r) {
                const size_t bytes = (size_t)g->ranks[r].B * g->ranks[r].W * sizeof(int16_t);
                if (!bytes) continue;
                nr = g->nccl.Send(g->ranks[r].send, bytes, ncclInt8, 0, g->comms[r], g->ranks[r].stream);      // rank r -> 0, on r's stream
                if (nr == ncclSuccess) nr = g->nccl.Recv(g->ranks[r].recv, bytes, ncclInt8, r, g->comms[0], root.stream);  // 0 <- r, on the root's
            }
            const ncclResult_t ne = g->nccl.GroupEnd();
            if (nr == ncclSuccess) nr = ne;
            if (nr != ncclSuccess) return fail(g, STN_ERR_DEVICE, std::string("stn_group_synthesize: RCCL gather: ") + g->nccl.GetErrorString(nr));
        } else {
            for (int r = 1; r < n; ++r) {  // rehearsal: the ranks share the root's GPU
                const size_t bytes = (size_t)g->ranks[r].B * g->ranks[r].W * sizeof(int16_t);
                if (!bytes) continue;
                HIPG(hipStreamWaitEvent(root.stream, g->ranks[r].done, 0));
                HIPG(hipMemcpyAsync(g->ranks[r].recv, g->ranks[r].send, bytes, hipMemcpyDeviceToDevice, root.stream));
            }
        }
        // ---- blocks -> pinned host memory, back to back; durations through the handles ---------------------------------------------
        size_t total = 0;
        for (Rank& k : g->ranks) total += (size_t)k.B * k.W * sizeof(int16_t);
        if (total > g->host_cap) {
            if (g->host_stage) HIPG(hipHostFree(g->host_stage));
            g->host_stage = nullptr; g->host_cap = 0;
            HIPG(hipHostMalloc(&g->host_stage, total + total / 4, hipHostMallocDefault));
            g->host_cap = total + total / 4;
        }
        size_t off = 0;
        for (int r = 0; r < n; ++r) {
            Rank& k = g->ranks[r];
            const size_t bytes = (size_t)k.B * k.W * sizeof(int16_t);
            if (bytes) HIPG(hipMemcpyAsync(static_cast<char*>(g->host_stage) + off, r < first_peer ? k.send : k.recv, bytes, hipMemcpyDeviceToHost, root.stream));
            off += bytes;
        }
        for (Rank& k : g->ranks) {
            k.dur.assign(k.B, 0.f);
            if (k.B && stn_batch_fetch(k.h, nullptr, 0, k.dur.data()) != STN_OK) return fail(g, STN_ERR_STATE, std::string("stn_group_synthesize: durations: ") + stn_last_error(k.h));
        }
        HIPG(hipStreamSynchronize(root.stream));
        for (int r = 1; r < n; ++r) { HIPG(hipSetDevice(g->ranks[r].device)); HIPG(hipStreamSynchronize(g->ranks[r].stream)); }
    } catch (const std::exception& e) { return fail(g, STN_ERR_DEVICE, std::string("stn_group_synthesize: ") + e.what()); }
    g->B = B;
    g->W = 0;
    for (Rank& k : g->ranks) g->W = std::max(g->W, k.W);
    if (samples_per_utt_out) *samples_per_utt_out = g->W;
    return STN_OK;
}

int stn_group_fetch_pcm16(stn_group* g, int16_t* pcm, size_t capacity_samples, float* duration) {
    if (!g) return STN_ERR_INVALID;
    if (g->B == 0) return fail(g, STN_ERR_STATE, "stn_group_fetch_pcm16: no synthesis to fetch");
    if (pcm && capacity_samples < (size_t)g->B * g->W) return fail(g, STN_ERR_INVALID, "stn_group_fetch_pcm16: capacity below B * samples_per_utt");
    const char* src = static_cast<const char*>(g->host_stage);
    for (Rank& k : g->ranks) {
        for (int j = 0; j < k.B; ++j) {
            const int i = k.idx[j];
            if (pcm) {
                int16_t* row = pcm + (size_t)i * g->W;
                std::memcpy(row, src + (size_t)j * k.W * sizeof(int16_t), (size_t)k.W * sizeof(int16_t));
                if (k.W < g->W) std::memset(row + k.W, 0, (size_t)(g->W - k.W) * sizeof(int16_t));
            }
            if (duration) duration[i] = k.dur[j];
        }
        src += (size_t)k.B * k.W * sizeof(int16_t);
    }
    return STN_OK;
}

int stn_group_last_shards(const stn_group* g, int32_t* rows_per_rank, int64_t* samples_per_rank) {
    if (!g) return STN_ERR_INVALID;
    for (size_t r = 0; r < g->ranks.size(); ++r) {
        if (rows_per_rank) rows_per_rank[r] = g->ranks[r].B;
        if (samples_per_rank) samples_per_rank[r] = g->ranks[r].W;
    }
    return STN_OK;
}

}  // extern "C"
