// Direct peer-store exchange of the slab-decomposed hot path (included by engine.hip inside its extern "C" block).
//
// The staged API (ofdft_dist_begin / _stage / _finish) leaves the all-to-alls to the host: six collectives and a dozen
// ctypes calls per evaluation from Python.  Here the library does the whole evaluation in ONE call and moves the data
// itself over the fully connected xGMI mesh of an MI355X node: every rank maps every peer's receive buffers and mailbox
// through hipIpc (handles travel once, at set-up, through whatever the host has: ofdft_ipc_export / ofdft_ipc_attach), then
//   * a stage's spectra go straight from the local send buffer into chunk [me] of each peer's receive buffer by ONE
//     kernel on the chain's stream whose workgroups store to all peers at once (blockIdx.y = peer: every direct link
//     me <-> p is driven concurrently -- P - 1 stream-ordered copies would use one link at a time);
//   * behind it a one-wave kernel stamps the evaluation's epoch into word [chain][me] of each peer's mailbox;
//   * the consumer's stream runs a one-wave kernel that waits (bounded) until every peer's word has reached the epoch.
// Two receive buffers per chain alternate, so a peer may deliver stage k + 1 while stage k is still being read.  The two
// small all-reduces (sum chi^2; the 13 energy sums) use the same mailboxes: every rank posts its numbers to every peer and
// sums the P contributions in rank order -- bitwise the same on all ranks.  One host synchronisation per evaluation.
// Nothing here is specific to separate GPUs: ranks sharing one GPU (tests) map each other's buffers the same way.
//
// Round 3: (a) the exchange is pipelined INSIDE a chain by kz chunks (engine_ctx.h: XchgChunks): each chain has a compute
// stream and a communication stream; chunk k's scatter + stamp run on the communication stream as soon as the chunk's y pass
// (or fused x pass) has finished, while the compute stream goes on with chunk k + 1; the consumer waits per chunk.
// (b) Failure handling is collective: epochs are derived from a per-call evaluation number (ranks stay in step after an
// error), a wait that runs out of patience (OFDFT_OPT_IPC_WAIT_MS, default 30 s) posts the evaluation number into every
// rank's abort word, every wait and the end of the call check that word -- all ranks fail the same evaluation.  (c) The arena
// must be fine-grained memory; there is no coarse-grained fallback (remote stores would not invalidate the local L2).
}  // extern "C"

struct ofdft_ipc_state {
    int P = 0, me = 0;
    // ONE allocation per rank (the arena) holds everything peers write into -- the four receive buffers (chain 0 a / b,
    // chain 1 a / b: objects 0..3) and the mailbox (object 4) -- so a rank opens exactly one hipIpc handle per peer
    // (handles of several small allocations of one process may name the same underlying block, which cannot be opened twice)
    void* arena = nullptr;
    size_t arena_bytes = 0, off[5] = {}, bytes[5] = {};
    void* peer_base[16] = {};          // rank p's arena in THIS process' address space (an opened handle)
    hipIpcMemHandle_t peer_handle[16] = {};
    // Arenas this rank has outgrown and mappings of arenas the peers have outgrown (round 5): kept until the context dies.  A
    // bigger term set used to FREE the arena while the peers still had it open through hipIpc, and the peers closed their mapping
    // and opened the replacement right after: one run in ~10 of four ranks failed there (hipIpcOpenMemHandle: invalid device
    // pointer).  Now nothing exported is freed and nothing opened is closed while the job runs: the replacement is a second
    // allocation (another address, another handle) and a second mapping.  (Sizing the arena once for the largest term set was
    // tried instead: at 512^3 on two ranks that is a 15-GB fine-grained allocation per rank, and mapping it did not return.)
    std::vector<void*> old_arenas, old_peer_bases;
    void* peer[5][16] = {};            // peer[w][p]: rank p's object w (p == me: the local pointer)
    // mailbox layout (this rank's copy): flags [3 kinds][16 ranks] u32 | sums [2 kinds][16 ranks][16] f64
    unsigned* flags = nullptr;
    double* sums = nullptr;
    unsigned eval_id = 0;              // evaluations started on this context (every rank counts the same calls): epochs derive from it
    unsigned ordinal[2] = {0, 0};      // exchanges (chunks) issued so far in this evaluation, per chain
    unsigned* d_stamp = nullptr;       // device scratch: the epoch values the flag copies read
    int* d_err = nullptr;              // [0] set by a wait that ran out of patience (1 + rank) or found the abort word (100), [1] abort word copy
    int* h_err = nullptr;
    hipStream_t side = nullptr;        // chain 1's compute stream
    hipStream_t comm[2] = {nullptr, nullptr};      // per chain: scatters + stamps (so that they overlap the chain's next kernels)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_cjoin[2] = {nullptr, nullptr};
    hipEvent_t ev_ready[2][16] = {};   // chunk k of chain c is complete in the send buffer
    hipEvent_t ev_sent[2][16] = {};    // ... and has been scattered (its local block [me] -> [me] included: the peers' stamps do not cover that one)
};

namespace {

constexpr size_t kIpcFlagWords = 3 * 16, kIpcSumDoubles = 2 * 16 * 16;
constexpr size_t kIpcMailboxBytes = kIpcFlagWords * sizeof(unsigned) + 64 + kIpcSumDoubles * sizeof(double);

void ipc_release(ofdft_ctx* c) {
    ofdft_ipc_state* s = c->ipc;
    if (!s) return;
    for (int p = 0; p < 16; ++p)
        if (s->peer_base[p]) (void)hipIpcCloseMemHandle(s->peer_base[p]);
    for (void* q : s->old_peer_bases) (void)hipIpcCloseMemHandle(q);
    for (void* q : s->old_arenas) (void)hipFree(q);
    for (const char* nm : {"x:recv0", "x:recv0b", "x:recv1", "x:recv1b"}) {        // the windows into the arena die with it
        auto it = c->ws.find(nm);
        if (it != c->ws.end() && it->second.borrowed) {
            c->ws_bytes -= it->second.bytes;
            c->ws.erase(it);
        }
    }
    if (s->arena) (void)hipFree(s->arena);
    if (s->d_stamp) (void)hipFree(s->d_stamp);
    if (s->d_err) (void)hipFree(s->d_err);
    if (s->h_err) (void)hipHostFree(s->h_err);
    if (s->side) (void)hipStreamDestroy(s->side);
    for (int ch = 0; ch < 2; ++ch) {
        if (s->comm[ch]) (void)hipStreamDestroy(s->comm[ch]);
        if (s->ev_cjoin[ch]) (void)hipEventDestroy(s->ev_cjoin[ch]);
        for (int k = 0; k < 16; ++k)
            if (s->ev_ready[ch][k]) (void)hipEventDestroy(s->ev_ready[ch][k]);
        for (int k = 0; k < 16; ++k)
            if (s->ev_sent[ch][k]) (void)hipEventDestroy(s->ev_sent[ch][k]);
    }
    if (s->ev_fork) (void)hipEventDestroy(s->ev_fork);
    if (s->ev_join) (void)hipEventDestroy(s->ev_join);
    delete s;
    c->ipc = nullptr;
}

struct IpcPeers { void* p[16]; };
// one wave: lane p waits until rank p's word has reached `epoch` (relaxed system-scope loads: the words are written by other
// ranks' kernels).  A lane gives up after `limit` ticks (100 MHz) -- or at once when some rank has already given up on THIS
// evaluation (the abort word of the local mailbox holds its number) -- and then tells everybody: the evaluation number goes
// into the abort word of every rank's mailbox, so that all ranks fail the same evaluation, the late one included.
__global__ void ipc_wait_kernel(const unsigned* flags, const unsigned* abort_word, IpcPeers mailbox, int P, int me, unsigned epoch,
                                unsigned eval_id, unsigned long long limit, int* err) {
    const int p = threadIdx.x;
    if (p >= P || p == me) return;
    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;      // an earlier wait of this evaluation
                                                                                              // already gave up: do not stack time limits
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();        // 100 MHz
    while ((int)(__hip_atomic_load(flags + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - epoch) < 0) {
        __builtin_amdgcn_s_sleep(32);
        // (monotone word: a rank that aborted evaluation N and N + 1 before a late rank reaches N must still fail N there)
        if ((int)(__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - eval_id) >= 0) {
            __hip_atomic_store(err, 100, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        if (__builtin_amdgcn_s_memrealtime() - t0 > limit) {
            __hip_atomic_store(err, 1 + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int q = 0; q < P; ++q)
                __hip_atomic_fetch_max(reinterpret_cast<unsigned*>(mailbox.p[q]) + 48, eval_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
        }
    }
    // system-scope acquire: what the peers stored before their stamps is what the kernels behind this one read
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
}
// end of an evaluation (after the last reduction, i.e. after every rank's last stamp): did anybody abort it?
__global__ void ipc_abort_check_kernel(const unsigned* abort_word, unsigned eval_id, int* err) {
    if (threadIdx.x == 0 && (int)(__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - eval_id) >= 0 &&
        __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
        __hip_atomic_store(err, 100, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// chunk [peer] of the local send buffer -> chunk [me] of rank peer's receive buffer (16-byte accesses; blockIdx.y = peer)
__global__ __launch_bounds__(256) void ipc_scatter_kernel(const u32x4* __restrict__ send, IpcPeers recv, int me, long long vec_per_peer) {
    const int peer = blockIdx.y;
    const u32x4* src = send + (long long)peer * vec_per_peer;
    u32x4* dst = reinterpret_cast<u32x4*>(recv.p[peer]) + (long long)me * vec_per_peer;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < vec_per_peer; i += (long long)gridDim.x * blockDim.x)
        __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
    // (No fence here.  A system-scope release per thread was tried: it is an L2 write-back per wave on this multi-XCD part --
    // the scatter kernels ran 10 x longer and dragged the kernels beside them along, 5.1 -> 22.5 ms per evaluation for two
    // ranks at 256^3.  The release that orders these stores before the stamp is ONE fence in the one-wave stamp kernel, which
    // runs behind this kernel on the same stream.)
}
// lane p: epoch -> word `word` of rank p's mailbox (system scope: another agent polls it)
__global__ void ipc_stamp_kernel(IpcPeers mailbox, int P, int me, int word, unsigned epoch) {
    const int p = threadIdx.x;
    // system-scope release: everything the kernels before this one (same stream: the scatter) stored is visible to the peers
    // before the stamp is -- one L2 write-back per chunk, not per scatter wave
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    if (p < P && p != me) __hip_atomic_store(reinterpret_cast<unsigned*>(mailbox.p[p]) + word, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// lane (p, i): this rank's i-th number -> its slot in rank p's mailbox
__global__ void ipc_post_kernel(IpcPeers mailbox, int P, size_t byte_off, const double* __restrict__ src, int n) {
    const int p = threadIdx.x / 16, i = threadIdx.x % 16;
    if (p < P && i < n) reinterpret_cast<double*>(reinterpret_cast<char*>(mailbox.p[p]) + byte_off)[i] = src[i];
}
// out[i] = sum over ranks (in rank order) of slots[p][i]
__global__ void ipc_sum_kernel(const double* slots, int P, int n, double* out) {
    const int i = threadIdx.x;
    if (i >= n) return;
    double t = 0.0;
    for (int p = 0; p < P; ++p) t += slots[p * 16 + i];
    out[i] = t;
}

int ipc_state(ofdft_ctx* c, ofdft_ipc_state** out) {
    if (!c->ipc) {
        ofdft_ipc_state* s = new (std::nothrow) ofdft_ipc_state();
        if (!s) return fail(c, OFDFT_ENOMEM, "out of host memory");
        s->P = c->nranks;
        s->me = c->rank;
        c->ipc = s;
        HIP_TRY(c, hipMalloc((void**)&s->d_stamp, 64));
        HIP_TRY(c, hipMalloc((void**)&s->d_err, sizeof(int)));
        HIP_TRY(c, hipMemset(s->d_err, 0, sizeof(int)));
        HIP_TRY(c, hipHostMalloc((void**)&s->h_err, sizeof(int)));
        HIP_TRY(c, hipStreamCreateWithFlags(&s->side, hipStreamNonBlocking));
        HIP_TRY(c, hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming));
        HIP_TRY(c, hipEventCreateWithFlags(&s->ev_join, hipEventDisableTiming));
        for (int ch = 0; ch < 2; ++ch) {
            HIP_TRY(c, hipStreamCreateWithFlags(&s->comm[ch], hipStreamNonBlocking));
            HIP_TRY(c, hipEventCreateWithFlags(&s->ev_cjoin[ch], hipEventDisableTiming));
            for (int k = 0; k < 16; ++k) HIP_TRY(c, hipEventCreateWithFlags(&s->ev_ready[ch][k], hipEventDisableTiming));
            for (int k = 0; k < 16; ++k) HIP_TRY(c, hipEventCreateWithFlags(&s->ev_sent[ch][k], hipEventDisableTiming));
        }
    }
    *out = c->ipc;
    return 0;
}

constexpr const char* kIpcRecvNames[4] = {"x:recv0", "x:recv0b", "x:recv1", "x:recv1b"};

// true when the exported arena still serves the active terms (the four receive workspaces are its windows, big enough)
bool ipc_arena_current(ofdft_ctx* c) {
    ofdft_ipc_state* s = c->ipc;
    if (!s || !s->arena) return false;
    for (int w = 0; w < 4; ++w) {
        auto it = c->ws.find(kIpcRecvNames[w]);
        if (it == c->ws.end() || !it->second.borrowed || it->second.p != (char*)s->arena + s->off[w]) return false;
        if (dist_buffer_bytes(c, w / 2) > s->bytes[w]) return false;
    }
    return true;
}

// (re)build the arena for the active terms and point the four receive-buffer workspaces into it
int ipc_arena(ofdft_ctx* c) {
    ofdft_ipc_state* s;
    if (int rc = ipc_state(c, &s)) return rc;
    if (ipc_arena_current(c)) return 0;
    // drop the old windows / arena (every peer has to attach again: ofdft_dist_closure checks that it did)
    for (int w = 0; w < 4; ++w) {
        auto it = c->ws.find(kIpcRecvNames[w]);
        if (it != c->ws.end()) {
            if (it->second.p && !it->second.borrowed) HIP_TRY(c, hipFree(it->second.p));
            c->ws_bytes -= it->second.bytes;
            c->ws.erase(it);
        }
    }
    if (s->arena) s->old_arenas.push_back(s->arena);       // (peers may still have it open: freed with the context)
    s->arena = nullptr;
    const size_t need[5] = {dist_buffer_bytes(c, 0), dist_buffer_bytes(c, 0), dist_buffer_bytes(c, 1), dist_buffer_bytes(c, 1),
                            kIpcMailboxBytes};
    size_t tot = 0;
    for (int w = 0; w < 5; ++w) {
        s->off[w] = tot;
        s->bytes[w] = need[w];
        tot += (need[w] + 4095) & ~(size_t)4095;
    }
    // fine-grained device memory: what a peer GPU stores here (spectra, epoch stamps) is coherent with this GPU's reads -- lines
    // of a coarse-grained allocation may linger in the local L2 across evaluations, which remote stores do not invalidate
    // -- so there is no fallback to an ordinary allocation: without fine-grained memory the transport is refused (the caller's
    // agreement step then takes the collective transport on every rank)
    if (hipError_t e = hipExtMallocWithFlags(&s->arena, tot, hipDeviceMallocFinegrained); e != hipSuccess) {
        (void)hipGetLastError();
        s->arena = nullptr;
        return fail(c, OFDFT_EHIP, "ipc transport: no fine-grained device memory for the exchange arena (%s): peers' stores would not be "
                                   "coherent with this GPU's reads -- use the collective transport", hipGetErrorString(e));
    }
    HIP_TRY(c, hipMemset((char*)s->arena + s->off[4], 0, kIpcMailboxBytes));
    HIP_TRY(c, hipDeviceSynchronize());
    s->arena_bytes = tot;
    for (int w = 0; w < 4; ++w) {
        DevBuf& b = c->ws[kIpcRecvNames[w]];
        b.p = (char*)s->arena + s->off[w];
        b.bytes = need[w];
        b.borrowed = true;
        c->ws_bytes += need[w];
        s->peer[w][s->me] = b.p;
    }
    s->peer[4][s->me] = (char*)s->arena + s->off[4];
    s->flags = (unsigned*)s->peer[4][s->me];
    s->sums = (double*)((char*)s->peer[4][s->me] + kIpcFlagWords * sizeof(unsigned) + 64);
    // (the epochs run on: a fresh mailbox holds zeros, which every later epoch exceeds)
    c->version++;                                       // captured graphs hold workspace addresses
    return 0;
}

// words of the local mailbox: flags [3 kinds][16 ranks] | abort word (first word of the 64-byte gap) | sums
inline const unsigned* ipc_abort_word(const ofdft_ipc_state* s) { return s->flags + kIpcFlagWords; }
inline unsigned long long ipc_limit_ticks(const ofdft_ctx* c) { return (unsigned long long)(c->ipc_wait_ms * 1e5); }   // 100 MHz

// one chunk of an exchange: `bytes_per_peer` bytes per peer from `send_region` (the chunk's region of the send buffer S[p ^ 1],
// [peer]-major) -> block [me] of the same region of every rank's receive buffer R[p ^ 1] (region_off bytes into it), the epoch
// stamp behind the data.  Runs on the chain's communication stream behind `ready` (recorded on the compute stream after the
// kernels that wrote the region).  Returns the epoch the consumers of the chunk wait for.
int ipc_send_chunk(ofdft_ctx* c, int chain, const cplx* send_region, size_t bytes_per_peer, size_t region_off, hipStream_t cs,
                   hipEvent_t ready, hipEvent_t sent, unsigned* epoch_out) {
    ofdft_ipc_state* s = c->ipc;
    const int w = 2 * chain + (c->recv_parity[chain] ^ 1);
    IpcPeers recv{}, mail{};
    for (int p = 0; p < s->P; ++p) {
        if (!s->peer[w][p]) return fail(c, OFDFT_ESTATE, "ipc transport: receive buffer %d of rank %d is not attached", w, p);
        recv.p[p] = (char*)s->peer[w][p] + region_off;
        mail.p[p] = s->peer[4][p];
    }
    if (bytes_per_peer % 16 || region_off % 16) return fail(c, OFDFT_EINVAL, "ipc transport: message size not a multiple of 16 bytes");
    hipStream_t ms = s->comm[chain];
    HIP_TRY(c, hipEventRecord(ready, cs));
    HIP_TRY(c, hipStreamWaitEvent(ms, ready, 0));
    const long long vec = (long long)(bytes_per_peer / 16);
    const int bx = (int)std::min<long long>(2048 / s->P + 1, (vec + 255) / 256);
    OFDFT_LAUNCH(c, ms, "ipc_scatter", ipc_scatter_kernel, dim3(bx, s->P), dim3(256), 0, (const u32x4*)send_region, recv, s->me, vec);
    const unsigned e = s->eval_id * 128u + (++s->ordinal[chain]);
    OFDFT_LAUNCH(c, ms, "ipc_sync", ipc_stamp_kernel, dim3(1), dim3(64), 0, mail, s->P, s->me, chain * 16 + s->me, e);
    HIP_TRY(c, hipEventRecord(sent, ms));
    *epoch_out = e;
    return 0;
}
// the consumer's side: the compute stream waits for this rank's own scatter of the chunk (it carries the local block) and
// then -- bounded -- until every peer's stamp of the chunk has arrived
int ipc_wait_chunk(ofdft_ctx* c, int chain, unsigned epoch, hipStream_t cs, hipEvent_t sent) {
    ofdft_ipc_state* s = c->ipc;
    HIP_TRY(c, hipStreamWaitEvent(cs, sent, 0));
    IpcPeers mail{};
    for (int p = 0; p < s->P; ++p) mail.p[p] = s->peer[4][p];
    OFDFT_LAUNCH(c, cs, "ipc_sync", ipc_wait_kernel, dim3(1), dim3(64), 0, (const unsigned*)(s->flags + chain * 16), ipc_abort_word(s), mail,
                 s->P, s->me, epoch, s->eval_id, ipc_limit_ticks(c), s->d_err);
    return 0;
}

// all ranks' `n` doubles at `src` (device) summed in rank order into `dst` (device); kind 0 / 1 = the two reductions
int ipc_allreduce(ofdft_ctx* c, int kind, const double* src, int n, double* dst, hipStream_t st) {
    ofdft_ipc_state* s = c->ipc;
    const size_t off = kIpcFlagWords * sizeof(unsigned) + 64 + sizeof(double) * ((size_t)kind * 256 + (size_t)s->me * 16);
    IpcPeers mail{};
    for (int p = 0; p < s->P; ++p) mail.p[p] = s->peer[4][p];
    OFDFT_LAUNCH(c, st, "ipc_sync", ipc_post_kernel, dim3(1), dim3(256), 0, mail, s->P, off, src, n);
    const unsigned e = s->eval_id * 4u + (unsigned)kind + 1u;
    OFDFT_LAUNCH(c, st, "ipc_sync", ipc_stamp_kernel, dim3(1), dim3(64), 0, mail, s->P, s->me, 2 * 16 + s->me, e);
    OFDFT_LAUNCH(c, st, "ipc_sync", ipc_wait_kernel, dim3(1), dim3(64), 0, (const unsigned*)(s->flags + 2 * 16), ipc_abort_word(s), mail, s->P,
                 s->me, e, s->eval_id, ipc_limit_ticks(c), s->d_err);
    OFDFT_LAUNCH(c, st, "ipc_sync", ipc_sum_kernel, dim3(1), dim3(64), 0, (const double*)(s->sums + (size_t)kind * 256), s->P, n, dst);
    return 0;
}

}  // namespace
extern "C" {

// hipIpc handle (64 bytes) of this rank's arena and the byte offsets of its five objects in it (0..3: the receive buffers
// chain 0 a / b, chain 1 a / b; 4: the mailbox).  Call after ofdft_set_terms (the buffers are sized by the active terms) and
// again -- on EVERY rank, followed by a new round of ofdft_ipc_attach -- when ofdft_set_terms changes them.
int ofdft_ipc_export(ofdft_ctx* c, void* handle64, unsigned long long* offsets5) {
    if (!c || !handle64 || !offsets5) return OFDFT_EINVAL;
    if (c->nranks < 2 || c->nranks > 16) return fail(c, OFDFT_EINVAL, "the ipc transport serves 2..16 ranks");
    if (!c->mask) return fail(c, OFDFT_ESTATE, "ofdft_set_terms has not been called");
    OFDFT_ON_DEVICE(c, c->device);
    if (int rc = ipc_arena(c)) return rc;
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "handle size");
    HIP_TRY(c, hipIpcGetMemHandle((hipIpcMemHandle_t*)handle64, c->ipc->arena));
    for (int w = 0; w < 5; ++w) offsets5[w] = c->ipc->off[w];
    return OFDFT_OK;
}

// map rank `peer`'s arena from what that rank exported (a handle already open here is kept)
int ofdft_ipc_attach(ofdft_ctx* c, int peer, const void* handle64, const unsigned long long* offsets5) {
    if (!c || !handle64 || !offsets5 || peer < 0 || peer >= c->nranks) return OFDFT_EINVAL;
    OFDFT_ON_DEVICE(c, c->device);
    ofdft_ipc_state* s;
    if (int rc = ipc_state(c, &s)) return rc;
    if (peer == s->me) return OFDFT_OK;
    hipIpcMemHandle_t h;
    std::memcpy(&h, handle64, sizeof(h));
    if (!s->peer_base[peer] || std::memcmp(&h, &s->peer_handle[peer], sizeof(h)) != 0) {
        for (int w = 0; w < 5; ++w) s->peer[w][peer] = nullptr;
        if (s->peer_base[peer]) {          // the peer has outgrown that arena: the mapping stays open until this context dies
            s->old_peer_bases.push_back(s->peer_base[peer]);
            s->peer_base[peer] = nullptr;
        }
        void* ptr = nullptr;
        HIP_TRY(c, hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess));
        s->peer_base[peer] = ptr;
        s->peer_handle[peer] = h;
        // prove the mapping with the runtime's own copies before any kernel stores through it (a copy that cannot reach the
        // peer returns an error; a kernel would fault): read a mailbox word, write a spare one (the last word of the 64-byte gap
        // between the flags and the sums)
        char* mb = (char*)ptr + offsets5[4];
        HIP_TRY(c, hipMemcpy(s->d_stamp, mb, sizeof(unsigned), hipMemcpyDeviceToDevice));
        HIP_TRY(c, hipMemcpy(mb + kIpcFlagWords * sizeof(unsigned) + 60, s->d_stamp + 1, sizeof(unsigned), hipMemcpyDeviceToDevice));
        HIP_TRY(c, hipDeviceSynchronize());
    }
    for (int w = 0; w < 5; ++w) s->peer[w][peer] = (char*)s->peer_base[peer] + offsets5[w];
    return OFDFT_OK;
}

// close every peer mapping (first step of a new export / attach round, before any rank frees its arena: include/ofdft_hip.h)
int ofdft_ipc_detach(ofdft_ctx* c) {
    if (!c) return OFDFT_EINVAL;
    ofdft_ipc_state* s = c->ipc;
    if (!s) return OFDFT_OK;
    OFDFT_ON_DEVICE(c, c->device);
    HIP_TRY(c, hipDeviceSynchronize());           // nothing of this rank's may still be storing through the mappings
    for (int p = 0; p < 16; ++p) {
        if (p == s->me) continue;
        for (int w = 0; w < 5; ++w) s->peer[w][p] = nullptr;
        if (s->peer_base[p]) {
            const hipError_t e = hipIpcCloseMemHandle(s->peer_base[p]);
            s->peer_base[p] = nullptr;
            std::memset(&s->peer_handle[p], 0, sizeof(s->peer_handle[p]));
            if (e != hipSuccess) {
                (void)hipGetLastError();
                return fail(c, OFDFT_EHIP, "ipc transport: closing the mapping of rank %d failed: %s", p, hipGetErrorString(e));
            }
        }
    }
    for (void* q : s->old_peer_bases) (void)hipIpcCloseMemHandle(q);
    s->old_peer_bases.clear();
    return OFDFT_OK;
}

// The closure chi -> (E, mu, chi.grad) on this rank's slab with the exchange done by the library (see the head of this
// file): every rank calls it with the same n_electrons.  Same results as the staged path (same kernels, same order).
int ofdft_dist_closure(ofdft_ctx* c, const void* chi_local, const void* vext_local, double n_electrons, double* E_terms,
                       double* mu_host, void* grad_local, void* v_work_local, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!c) return OFDFT_EINVAL;
    OFDFT_ON_DEVICE(c, c->device);
    if (int rc = begin_call(c, st)) return rc;
    if (!chi_local || !E_terms || !grad_local || !v_work_local) return fail(c, OFDFT_EINVAL, "null argument");
    if (!c->mask) return fail(c, OFDFT_ESTATE, "ofdft_set_terms has not been called");
    if (c->nranks < 2 || !c->ipc) return fail(c, OFDFT_ESTATE, "ofdft_dist_closure needs a slab-decomposed context with the ipc transport attached");
    if ((c->mask & OFDFT_ION_ELECTRON) && !vext_local) return fail(c, OFDFT_EINVAL, "IonElectron term needs vext");
    if (gga_needs_laplacian(c) && !c->gga_split) return fail(c, OFDFT_EINVAL, "Laplacian-dependent Pauli-Gaussian members need OFDFT_OPT_GGA_SPLIT = 1");
    if (wts_active(c)) return fail(c, OFDFT_EINVAL, "the stabilised Wang-Teter style functional is served by single-GPU contexts");
    ofdft_ipc_state* s = c->ipc;
    for (int w = 0; w < 5; ++w)
        for (int p = 0; p < s->P; ++p)
            if (!s->peer[w][p]) return fail(c, OFDFT_ESTATE, "ipc transport: object %d of rank %d is not attached (ofdft_ipc_export / ofdft_ipc_attach)", w, p);
    if (!ipc_arena_current(c))          // a set_terms that needs bigger buffers than the exported arena invalidates the peers' mappings
        return fail(c, OFDFT_ESTATE, "ipc transport: exchange buffers changed since ofdft_ipc_export (export and attach again on every rank)");
    const real* chi = (const real*)chi_local;
    // a new evaluation: its number (the same on every rank) seeds every epoch; parities and ordinals start over, so ranks are in
    // step again even after an evaluation that failed half-way
    s->eval_id++;
    s->ordinal[0] = s->ordinal[1] = 0;
    c->recv_parity[0] = c->recv_parity[1] = 0;
    HIP_TRY(c, hipMemsetAsync(s->d_err, 0, sizeof(int), st));
    // ---- sum chi^2 over all ranks -> closure scale on the device
    const int blocks = grid_for(c->npts / 2 + 1, kRedThreads, kRedBlocks);
    OFDFT_LAUNCH(c, st, "sum", (sum_kernel<true>), dim3(blocks), dim3(kRedThreads), 0, chi, c->npts, c->d_partial);
    OFDFT_REDUCE(c, st, c->d_partial, blocks, 1, c->d_reduced + kSumsqSlot);
    if (int rc = ipc_allreduce(c, 0, c->d_reduced + kSumsqSlot, 1, c->d_reduced + kSumsqSlot, st)) return rc;
    OFDFT_LAUNCH(c, st, "reduce", closure_scale_kernel, dim3(1), dim3(64), 0, c->d_reduced + kSumsqSlot, c->d_scal, n_electrons,
                 c->vol / (double)c->npts_g);
    ZRun& r = zrun(c);
    r.ds = DenSrc{chi, 0.0, 1, c->d_scal};
    r.nel = n_electrons;
    r.vext = (const real*)vext_local;
    r.v_out = (real*)v_work_local;
    r.stage[0] = r.stage[1] = 0;
    r.deferred.clear();
    r.forked = false;
    r.closure = false;
    r.vpart_deferred = false;
    r.za.v_part_deferred = 0;
    r.xlist[0].clear();
    r.xlist[1].clear();
    // ---- the two chains on their own compute streams, each with a communication stream: the scatter of chunk k is enqueued
    // behind the kernels that produced it and runs while the compute stream works on chunk k + 1; consumers wait per chunk
    HIP_TRY(c, hipEventRecord(s->ev_fork, st));
    HIP_TRY(c, hipStreamWaitEvent(s->side, s->ev_fork, 0));
    const int K = c->xc.n;
    unsigned pend[2][16] = {};                  // epoch of the chunk delivery the chain's next consuming step waits for (0: none)
    for (int step = 1; step <= 6; ++step)
        for (int chain = 0; chain < 2; ++chain) {
            hipStream_t cs = chain == 0 ? st : s->side;
            const bool sends = step == 1 || step == 2 || step == 4 || step == 5;
            bool sent = false;
            for (int k = 0; k < K; ++k) {
                if (pend[chain][k]) {           // chunk k of the exchange this step consumes
                    if (int rc = ipc_wait_chunk(c, chain, pend[chain][k], cs, s->ev_sent[chain][k])) return rc;
                    pend[chain][k] = 0;
                }
                int rc;
                switch (step) {
                    case 1: rc = zstage1(c, cs, chain, k); break;
                    case 2: rc = zstage2(c, cs, chain, k); break;
                    case 3: rc = zstage3(c, cs, chain, 1, k); break;
                    case 4: rc = zstage3(c, cs, chain, 2, k); break;
                    case 5: rc = zstage4(c, cs, chain, k); break;
                    default: rc = chain == 0 ? zstage5(c, nullptr, cs, false, 1, k) : 0; break;     // (the divergence belongs to chain 0)
                }
                if (rc) return rc;
                if (sends && !r.xlist[chain].empty()) {
                    cplx *send, *recv;
                    if ((rc = dist_buffers(c, chain, &send, &recv))) return rc;
                    const XcView v = xc_view(c, k);
                    const size_t narr = r.xlist[chain].size();
                    const size_t bytes = sizeof(cplx) * (size_t)c->xg.nxl * (size_t)v.arr_sz * narr;
                    if ((rc = ipc_send_chunk(c, chain, send + narr * v.base1, bytes, sizeof(cplx) * narr * (size_t)v.base1, cs,
                                             s->ev_ready[chain][k], s->ev_sent[chain][k], &pend[chain][k])))
                        return rc;
                    sent = true;
                }
            }
            if (sent) c->recv_parity[chain] ^= 1;      // the chain's next step reads what was just sent
        }
    // the divergence (chain 0, step 6) has been y-inverted chunk by chunk; join the nonlocal chain and the communication streams
    HIP_TRY(c, hipEventRecord(s->ev_join, s->side));
    HIP_TRY(c, hipStreamWaitEvent(st, s->ev_join, 0));
    for (int ch = 0; ch < 2; ++ch) {
        HIP_TRY(c, hipEventRecord(s->ev_cjoin[ch], s->comm[ch]));
        HIP_TRY(c, hipStreamWaitEvent(st, s->ev_cjoin[ch], 0));
    }
    if (int rc = zstage5(c, nullptr, st, false, 2)) return rc;     // combine; local sums -> d_reduced[0..12]
    if (int rc = ipc_allreduce(c, 1, c->d_reduced, kNSums + 1, c->d_reduced, st)) return rc;
    OFDFT_LAUNCH(c, st, "chi_grad", chi_grad_kernel, dim3(grid_for(c->npts / 2 + 1)), dim3(256), 0, chi, (const real*)v_work_local,
                 (real*)grad_local, c->npts, 0.0, (const acc_t*)c->d_scal, 2.0 * c->dV, 0.0, (const acc_t*)(c->d_reduced + 8), c->dV,
                 n_electrons);
    HIP_TRY(c, hipMemcpyAsync(c->h_partial, c->d_reduced, sizeof(double) * (kNSums + 1), hipMemcpyDeviceToHost, st));
    OFDFT_LAUNCH(c, st, "ipc_sync", ipc_abort_check_kernel, dim3(1), dim3(64), 0, ipc_abort_word(s), s->eval_id, s->d_err);
    HIP_TRY(c, hipMemcpyAsync(s->h_err, s->d_err, sizeof(int), hipMemcpyDeviceToHost, st));
    if (int rc = end_call(c, st)) return rc;
    if (*s->h_err) {        // every rank reaches this for the same evaluation (abort word); the next call starts from a clean slate
        const int code = *s->h_err;
        if (code == 100)
            return fail(c, OFDFT_EHIP, "ipc transport: evaluation %u was aborted by another rank (a delivery wait ran out of patience there)", s->eval_id);
        return fail(c, OFDFT_EHIP, "ipc transport: no delivery from rank %d within %.0f ms (OFDFT_OPT_IPC_WAIT_MS); evaluation %u aborted on all ranks",
                    code - 1, c->ipc_wait_ms, s->eval_id);
    }
    double vn;
    for (int i = 0; i < OFDFT_NTERMS; ++i) E_terms[i] = 0.0;
    energies_from_sums(c, c->h_partial, c->h_partial + kCombineScalars, E_terms, &vn);
    if (mu_host) *mu_host = vn / n_electrons;
    return OFDFT_OK;
}
