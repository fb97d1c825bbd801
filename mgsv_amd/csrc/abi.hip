// ABI bookkeeping: version, thread-local error text, device query.
#include "common.h"

#include <stdarg.h>
#include <string.h>

static thread_local char g_err[512] = "";

void made_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int made_abi_version(void) { return MADE_ABI_VERSION; }

// f32 products: 0 = exact (v_mfma_f32_32x32x2_f32), 1 = split-bf16 (three bf16 products per f32 product: common.h).  Process-wide; read at launch.
int g_made_f32_products = 0;
extern "C" int made_set_f32_products(int mode) {
    if (mode != 0 && mode != 1) { made_set_error("made_set_f32_products: mode %d not in {0, 1}", mode); return MADE_ERR_INVALID_ARG; }
    g_made_f32_products = mode;
    return MADE_OK;
}
extern "C" int made_get_f32_products(void) { return g_made_f32_products; }

extern "C" const char* made_last_error(void) { return g_err; }

extern "C" int made_device_info(char* name, int name_len, int* cu_count, int* is_gfx950) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) {
        made_set_error("made_device_info: no HIP device: %s", hipGetErrorString(e));
        return MADE_ERR_HIP;
    }
    hipDeviceProp_t p;
    e = hipGetDeviceProperties(&p, dev);
    if (e != hipSuccess) {
        made_set_error("made_device_info: %s", hipGetErrorString(e));
        return MADE_ERR_HIP;
    }
    if (name && name_len > 0) {
        strncpy(name, p.gcnArchName, (size_t)name_len - 1);
        name[name_len - 1] = 0;
    }
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (is_gfx950) *is_gfx950 = strncmp(p.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
    return MADE_OK;
}


// =================================================================================================
// Launch tape: a recorded sequence of the library's kernel launches (with the stream-to-stream dependencies, memsets and copies
// between them) that one call replays from a C loop.  The training step is ~600 launches of 5-50 us: issued from Python (10 us of
// interpreter + ctypes per launch) or through hipGraphLaunch (which walks its nodes on the host at ~9 us apiece on this ROCm) the
// host, not the GPU, sets the pace of the decoder's chain of small launches; replayed from here a launch costs the runtime's own
// 2-3 us and both streams keep their overlap.  Everything that changes between replays must live in device memory (the batch
// buffers, MadeDropout.seed_device, MadeAdamDeviceState): the tape holds argument BYTES.
#include <vector>

namespace {
enum { TAPE_KERNEL = 0, TAPE_WAIT = 1, TAPE_MEMSET = 2, TAPE_COPY = 3, TAPE_EV_RECORD = 4, TAPE_EV_WAIT = 5, TAPE_CALLBACK = 6 };
struct TapeOp {
    int kind;
    const void* fn; dim3 grid, block; unsigned lds; hipStream_t st, st2;
    size_t arg_off; int nargs; size_t ptr_off;               // kernel: argument bytes in `blob`, offsets of the arguments in `offs`
    hipEvent_t ev;                                           // wait: recorded on st, awaited by st2
    void* dst; const void* src; size_t nbytes; int value;    // memset / copy
    MadeTapeCallback cb; void* cb_user;                      // host callback (made_tape_callback)
};
struct Tape {
    std::vector<TapeOp> ops;
    std::vector<unsigned char> blob;
    std::vector<size_t> offs;
    std::vector<void*> ptrs;                                 // scratch of replay
    std::vector<hipEvent_t> slots;                           // events of made_tape_event, by slot
};
}  // namespace

thread_local void* g_made_tape = nullptr;

void made_tape_push_kernel(const void* fn, dim3 grid, dim3 block, unsigned lds, hipStream_t st, void** arg_ptrs, const size_t* arg_sizes, int n) {
    Tape* t = (Tape*)g_made_tape;
    TapeOp op{};
    op.kind = TAPE_KERNEL; op.fn = fn; op.grid = grid; op.block = block; op.lds = lds; op.st = st;
    op.nargs = n; op.ptr_off = t->offs.size();
    for (int i = 0; i < n; ++i) {
        size_t off = (t->blob.size() + 15) & ~(size_t)15;
        t->blob.resize(off + arg_sizes[i]);
        memcpy(t->blob.data() + off, arg_ptrs[i], arg_sizes[i]);
        t->offs.push_back(off);
    }
    t->ops.push_back(op);
}

extern "C" int made_tape_begin(void) {
    MADE_REQUIRE(g_made_tape == nullptr, "made_tape_begin: this thread is already recording");
    g_made_tape = new Tape();
    return MADE_OK;
}

// A program asks for the same cross-stream dependency more than once (two helpers in a row each make the second stream wait for the first):
// every record + wait costs the waiting stream >= 6 us on this stack even when there is nothing to wait for (tools/probes/stream_value_probe.hip).
// Dropped here, once, at the end of the recording:
//   * a record on a stream that has been given nothing since its previous record marks the same point: its event becomes an alias of the earlier one;
//   * a wait of a stream for an event it has already waited for (same record) adds nothing.
static void tape_dedup(Tape* t) {
    const size_t n = t->ops.size();
    std::vector<std::pair<hipEvent_t, hipEvent_t>> alias;                 // event -> the earlier event that marks the same point
    auto canon = [&](hipEvent_t e) { for (auto& a : alias) if (a.first == e) return a.second; return e; };
    std::vector<std::pair<hipStream_t, hipEvent_t>> last_rec;             // stream -> event of its latest record with nothing issued to the stream since
    std::vector<std::pair<hipStream_t, std::vector<hipEvent_t>>> waited;  // stream -> events it has waited for
    auto forget = [&](hipStream_t st) { for (auto& lr : last_rec) if (lr.first == st) lr.second = nullptr; };
    std::vector<char> drop(n, 0);
    for (size_t i = 0; i < n; ++i) {
        TapeOp& op = t->ops[i];
        if (op.kind == TAPE_EV_RECORD) {
            hipEvent_t prev = nullptr;
            for (auto& lr : last_rec) if (lr.first == op.st) prev = lr.second;
            if (prev != nullptr) { alias.push_back({op.ev, prev}); drop[i] = 1; continue; }
            bool found = false;
            for (auto& lr : last_rec) if (lr.first == op.st) { lr.second = op.ev; found = true; }
            if (!found) last_rec.push_back({op.st, op.ev});
            for (auto& wd : waited) {                         // (an event recorded again: a later wait for it is a new dependency)
                auto& v = wd.second;
                for (size_t k = 0; k < v.size();) { if (v[k] == op.ev) v.erase(v.begin() + (long)k); else ++k; }
            }
        } else if (op.kind == TAPE_EV_WAIT) {
            op.ev = canon(op.ev);
            std::vector<hipEvent_t>* w = nullptr;
            for (auto& wd : waited) if (wd.first == op.st) w = &wd.second;
            if (w == nullptr) { waited.push_back({op.st, {}}); w = &waited.back().second; }
            bool seen = false;
            for (hipEvent_t e : *w) seen = seen || e == op.ev;
            if (seen) { drop[i] = 1; continue; }
            w->push_back(op.ev);
            forget(op.st);                                    // (the stream's next record covers what it has just waited for as well)
        } else if (op.kind == TAPE_CALLBACK) {
            for (auto& lr : last_rec) lr.second = nullptr;    // (a host callback may put work of the framework on any stream)
            for (auto& wd : waited) wd.second.clear();
        } else if (op.kind == TAPE_WAIT) {
            forget(op.st); forget(op.st2);
        } else {
            forget(op.st);
        }
    }
    std::vector<TapeOp> out;
    out.reserve(n);
    for (size_t i = 0; i < n; ++i) if (!drop[i]) out.push_back(t->ops[i]);
    t->ops.swap(out);
}

extern "C" int made_tape_end(uint64_t* handle) {
    MADE_REQUIRE(g_made_tape != nullptr && handle != nullptr, "made_tape_end: not recording");
    if (made_variant_env("MADE_TAPE_NO_DEDUP") == nullptr) tape_dedup((Tape*)g_made_tape);
    *handle = (uint64_t)(uintptr_t)g_made_tape;
    g_made_tape = nullptr;
    return MADE_OK;
}

extern "C" int made_tape_free(uint64_t handle) {
    Tape* t = (Tape*)(uintptr_t)handle;
    if (t == nullptr) return MADE_OK;
    for (auto& op : t->ops)
        if (op.kind == TAPE_WAIT && op.ev) (void)hipEventDestroy(op.ev);
    for (auto ev : t->slots)
        if (ev) (void)hipEventDestroy(ev);
    delete t;
    return MADE_OK;
}

extern "C" int made_tape_count(uint64_t handle, int64_t* kernels, int64_t* waits, int64_t* others) {
    Tape* t = (Tape*)(uintptr_t)handle;
    MADE_REQUIRE(t != nullptr, "made_tape_count: null tape");
    int64_t k = 0, w = 0, o = 0;
    for (auto& op : t->ops) { if (op.kind == TAPE_KERNEL) ++k; else if (op.kind == TAPE_WAIT) ++w; else ++o; }
    if (kernels) *kernels = k;
    if (waits) *waits = w;
    if (others) *others = o;
    return MADE_OK;
}

// dst_stream waits for everything src_stream has been given so far (torch's Stream.wait_stream / Event.record + wait_event)
extern "C" int made_stream_wait(void* src_stream, void* dst_stream) {
    hipEvent_t ev = nullptr;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { made_set_error("made_stream_wait: hipEventCreate failed"); return MADE_ERR_HIP; }
    hipError_t e = hipEventRecord(ev, (hipStream_t)src_stream);
    if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)dst_stream, ev, 0);
    if (e != hipSuccess) { (void)hipEventDestroy(ev); made_set_error("made_stream_wait: %s", hipGetErrorString(e)); return MADE_ERR_HIP; }
    if (g_made_tape) {
        TapeOp op{};
        op.kind = TAPE_WAIT; op.st = (hipStream_t)src_stream; op.st2 = (hipStream_t)dst_stream; op.ev = ev;
        ((Tape*)g_made_tape)->ops.push_back(op);
    } else {
        (void)hipEventDestroy(ev);                           // (destruction is deferred by the runtime until the event has completed)
    }
    return MADE_OK;
}

// Recording only (nothing is executed: the caller's own event call does that): "record event `slot` on `stream`" (op 0) or "`stream`
// waits for event `slot`" (op 1) -- the framework's Event.record / Stream.wait_event pair, mirrored into the tape.
extern "C" int made_tape_event(int32_t op_kind, int32_t slot, void* stream) {
    Tape* t = (Tape*)g_made_tape;
    MADE_REQUIRE(t != nullptr, "made_tape_event: not recording");
    MADE_REQUIRE((op_kind == 0 || op_kind == 1) && slot >= 0 && slot < 65536, "made_tape_event: bad op / slot");
    if ((size_t)slot >= t->slots.size()) t->slots.resize((size_t)slot + 1, nullptr);
    if (t->slots[slot] == nullptr) {
        MADE_REQUIRE(op_kind == 0, "made_tape_event: slot %d is waited for before it was recorded", slot);
        if (hipEventCreateWithFlags(&t->slots[slot], hipEventDisableTiming) != hipSuccess) { made_set_error("made_tape_event: hipEventCreate failed"); return MADE_ERR_HIP; }
    }
    TapeOp op{};
    op.kind = op_kind == 0 ? TAPE_EV_RECORD : TAPE_EV_WAIT; op.st = (hipStream_t)stream; op.ev = t->slots[slot];
    t->ops.push_back(op);
    return MADE_OK;
}

// Recording only: a HOST callback at this point of the issue order (the caller performs the action itself while recording).  What the
// data-parallel training step uses it for: the gradient all-reduces are RCCL calls of the framework (their own stream, ordered against
// the step's streams by events the framework records at call time), so a replay has to make the same calls at the same points --
// after everything in front of them has been ISSUED, before anything behind them is.
extern "C" int made_tape_callback(MadeTapeCallback fn, void* user) {
    Tape* t = (Tape*)g_made_tape;
    MADE_REQUIRE(t != nullptr, "made_tape_callback: not recording");
    MADE_REQUIRE(fn != nullptr, "made_tape_callback: null function");
    TapeOp op{};
    op.kind = TAPE_CALLBACK; op.cb = fn; op.cb_user = user;
    t->ops.push_back(op);
    return MADE_OK;
}

extern "C" int made_memset_async(void* dst, int32_t value, int64_t nbytes, void* stream) {
    MADE_REQUIRE(dst != nullptr && nbytes >= 0, "made_memset_async: bad arguments");
    if (nbytes == 0) return MADE_OK;
    hipError_t e = hipMemsetAsync(dst, value, (size_t)nbytes, (hipStream_t)stream);
    if (e != hipSuccess) { made_set_error("made_memset_async: %s", hipGetErrorString(e)); return MADE_ERR_HIP; }
    if (g_made_tape) {
        TapeOp op{};
        op.kind = TAPE_MEMSET; op.st = (hipStream_t)stream; op.dst = dst; op.nbytes = (size_t)nbytes; op.value = value;
        ((Tape*)g_made_tape)->ops.push_back(op);
    }
    return MADE_OK;
}

// a few 32-bit words handed over as KERNEL ARGUMENTS (stream-ordered, nothing for the host to keep alive): the per-step scalars of a
// replayed training step -- dropout seed, learning rates, the device-side step count (mgsv_amd/trainer.py TrainStepGraph.step)
struct StoreWords { uint32_t w[4]; };
__global__ void store_words_kernel(uint32_t* dst, StoreWords v, int n) {
    if ((int)threadIdx.x < n) dst[threadIdx.x] = v.w[threadIdx.x];
}
extern "C" int made_store_words(void* dst, const uint32_t* words, int32_t n_words, void* stream) {
    MADE_REQUIRE(dst != nullptr && words != nullptr && n_words >= 1 && n_words <= 4 && ((uintptr_t)dst & 3) == 0, "made_store_words: bad arguments");
    StoreWords v{};
    for (int i = 0; i < n_words; ++i) v.w[i] = words[i];
    hipLaunchKernelGGL(store_words_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (uint32_t*)dst, v, (int)n_words);
    return made_check_launch("made_store_words");
}

extern "C" int made_copy_async(void* dst, const void* src, int64_t nbytes, void* stream) {
    MADE_REQUIRE(dst != nullptr && src != nullptr && nbytes >= 0, "made_copy_async: bad arguments");
    if (nbytes == 0) return MADE_OK;
    hipError_t e = hipMemcpyAsync(dst, src, (size_t)nbytes, hipMemcpyDeviceToDevice, (hipStream_t)stream);
    if (e != hipSuccess) { made_set_error("made_copy_async: %s", hipGetErrorString(e)); return MADE_ERR_HIP; }
    if (g_made_tape) {
        TapeOp op{};
        op.kind = TAPE_COPY; op.st = (hipStream_t)stream; op.dst = dst; op.src = src; op.nbytes = (size_t)nbytes;
        ((Tape*)g_made_tape)->ops.push_back(op);
    }
    return MADE_OK;
}

// Re-order the tape's ISSUE order (the order made_tape_replay walks it in) without touching what the GPU may observe: every stream
// keeps its own order, a "stream waits for event" is issued after the "record" it refers to, and operations on one event keep their
// recorded order.  Why: the recording is in program order, and a program issues a whole side-stream branch (tens of launches) before
// it returns to the main stream -- at ~5 us of host time per launch the main stream then sits idle behind its last launch (measured:
// 0.87 ms of a 6.2 ms step).  Here the streams' queues are drained round-robin, `main_weight` operations of the busiest stream for
// every one of each other stream, so all streams are fed at the same time.
extern "C" int made_tape_interleave(uint64_t handle, int32_t main_weight) {
    Tape* t = (Tape*)(uintptr_t)handle;
    MADE_REQUIRE(t != nullptr, "made_tape_interleave: null tape");
    MADE_REQUIRE(main_weight >= 1 && main_weight <= 16, "made_tape_interleave: main_weight out of [1, 16]");
    const size_t n = t->ops.size();
    // a host callback is a point of the ISSUE order: nothing moves across it.  Tapes with callbacks are re-ordered segment by segment.
    {
        bool has_cb = false;
        for (size_t i = 0; i < n; ++i) has_cb = has_cb || t->ops[i].kind == TAPE_CALLBACK;
        if (has_cb) {
            for (size_t i = 0; i < n; ++i) if (t->ops[i].kind == TAPE_WAIT) return MADE_OK;
            std::vector<TapeOp> all;
            all.swap(t->ops);
            std::vector<TapeOp> out;
            out.reserve(n);
            size_t b = 0;
            int rc = MADE_OK;
            while (b <= n && rc == MADE_OK) {
                size_t e = b;
                while (e < n && all[e].kind != TAPE_CALLBACK) ++e;
                t->ops.assign(all.begin() + (long)b, all.begin() + (long)e);
                if (t->ops.size() > 1) rc = made_tape_interleave(handle, main_weight);
                out.insert(out.end(), t->ops.begin(), t->ops.end());
                if (e < n) out.push_back(all[e]);
                b = e + 1;
            }
            if (rc != MADE_OK) { t->ops.swap(all); return rc; }
            t->ops.swap(out);
            return MADE_OK;
        }
    }
    // streams in order of first appearance; (a TAPE_WAIT op -- made_stream_wait -- belongs to both of its streams: keep such tapes as they are)
    std::vector<hipStream_t> streams;
    std::vector<int> sid(n);
    for (size_t i = 0; i < n; ++i) {
        if (t->ops[i].kind == TAPE_WAIT) return MADE_OK;
        size_t k = 0;
        while (k < streams.size() && streams[k] != t->ops[i].st) ++k;
        if (k == streams.size()) streams.push_back(t->ops[i].st);
        sid[i] = (int)k;
    }
    const size_t ns = streams.size();
    if (ns < 2) return MADE_OK;
    // dependency of op i on an earlier op of ANOTHER stream: the previous operation on the same event
    std::vector<long> dep(n, -1);
    {
        std::vector<std::pair<hipEvent_t, long>> last;                // event -> index of the latest op touching it
        for (size_t i = 0; i < n; ++i) {
            const TapeOp& op = t->ops[i];
            if (op.kind != TAPE_EV_RECORD && op.kind != TAPE_EV_WAIT) continue;
            size_t k = 0;
            while (k < last.size() && last[k].first != op.ev) ++k;
            if (k == last.size()) last.push_back({op.ev, -1});
            dep[i] = last[k].second;
            last[k].second = (long)i;
        }
    }
    std::vector<std::vector<size_t>> q(ns);
    for (size_t i = 0; i < n; ++i) q[sid[i]].push_back(i);
    size_t main_s = 0;
    for (size_t k = 1; k < ns; ++k) if (q[k].size() > q[main_s].size()) main_s = k;
    std::vector<size_t> head(ns, 0);
    std::vector<char> done(n, 0);
    std::vector<size_t> order;
    order.reserve(n);
    auto ready = [&](size_t k) { return head[k] < q[k].size() && (dep[q[k][head[k]]] < 0 || done[(size_t)dep[q[k][head[k]]]]); };
    while (order.size() < n) {
        bool progressed = false;
        for (size_t k = 0; k < ns; ++k) {
            const int quota = k == main_s ? main_weight : 1;
            for (int c = 0; c < quota && ready(k); ++c) {
                const size_t i = q[k][head[k]++];
                done[i] = 1; order.push_back(i); progressed = true;
            }
        }
        if (!progressed) { made_set_error("made_tape_interleave: the tape's dependencies do not resolve (a wait without its record)"); return MADE_ERR_INVALID_ARG; }
    }
    std::vector<TapeOp> re;
    re.reserve(n);
    for (size_t i : order) re.push_back(t->ops[i]);
    t->ops.swap(re);
    return MADE_OK;
}

extern "C" int made_tape_op(uint64_t handle, int64_t index, int32_t* kind, uint64_t* function, uint64_t* stream, uint32_t* grid3) {
    Tape* t = (Tape*)(uintptr_t)handle;
    MADE_REQUIRE(t != nullptr && index >= 0 && (size_t)index < t->ops.size(), "made_tape_op: bad tape or index");
    const TapeOp& op = t->ops[(size_t)index];
    if (kind) *kind = (int32_t)op.kind;
    if (function) *function = (uint64_t)(uintptr_t)op.fn;
    if (stream) *stream = (uint64_t)(uintptr_t)op.st;
    if (grid3) { grid3[0] = op.grid.x; grid3[1] = op.grid.y; grid3[2] = op.grid.z; }
    return MADE_OK;
}

static int tape_replay_range(Tape* t, size_t first, size_t count);

extern "C" int made_tape_replay_range(uint64_t handle, int64_t first, int64_t count) {
    Tape* t = (Tape*)(uintptr_t)handle;
    MADE_REQUIRE(t != nullptr, "made_tape_replay_range: null tape");
    MADE_REQUIRE(g_made_tape == nullptr, "made_tape_replay_range: this thread is recording");
    MADE_REQUIRE(first >= 0 && count >= 0 && (size_t)(first + count) <= t->ops.size(), "made_tape_replay_range: [%lld, +%lld) outside the tape's %zu operations",
                 (long long)first, (long long)count, t->ops.size());
    return tape_replay_range(t, (size_t)first, (size_t)count);
}

extern "C" int made_tape_replay(uint64_t handle) {
    Tape* t = (Tape*)(uintptr_t)handle;
    MADE_REQUIRE(t != nullptr, "made_tape_replay: null tape");
    MADE_REQUIRE(g_made_tape == nullptr, "made_tape_replay: this thread is recording");
    return tape_replay_range(t, 0, t->ops.size());
}

static int tape_replay_range(Tape* t, size_t first, size_t count) {
    unsigned char* blob = t->blob.data();
    for (size_t oi = first; oi < first + count; ++oi) {
        TapeOp& op = t->ops[oi];
        hipError_t e = hipSuccess;
        switch (op.kind) {
            case TAPE_KERNEL: {
                t->ptrs.resize((size_t)op.nargs > t->ptrs.size() ? (size_t)op.nargs : t->ptrs.size());
                for (int i = 0; i < op.nargs; ++i) t->ptrs[i] = blob + t->offs[op.ptr_off + i];
                e = hipLaunchKernel(op.fn, op.grid, op.block, t->ptrs.data(), op.lds, op.st);
                break;
            }
            case TAPE_WAIT:
                e = hipEventRecord(op.ev, op.st);
                if (e == hipSuccess) e = hipStreamWaitEvent(op.st2, op.ev, 0);
                break;
            case TAPE_EV_RECORD: e = hipEventRecord(op.ev, op.st); break;
            case TAPE_EV_WAIT: e = hipStreamWaitEvent(op.st, op.ev, 0); break;
            case TAPE_MEMSET: e = hipMemsetAsync(op.dst, op.value, op.nbytes, op.st); break;
            case TAPE_CALLBACK: {
                const int rc = op.cb(op.cb_user);
                if (rc != 0) { made_set_error("made_tape_replay: the host callback of operation %zu returned %d", oi, rc); return MADE_ERR_INVALID_ARG; }
                break;
            }
            default: e = hipMemcpyAsync(op.dst, op.src, op.nbytes, hipMemcpyDeviceToDevice, op.st); break;
        }
        if (e != hipSuccess) { made_set_error("made_tape_replay: %s", hipGetErrorString(e)); return MADE_ERR_HIP; }
    }
    return MADE_OK;
}
