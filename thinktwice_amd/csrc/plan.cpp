// Launch plans: the C-side executor of the forward path (SURVEY 8b B3 -- tt_encoder_fwd / tt_decoder_fwd).
//
// The reference sequences its forward in Python (encoder_decoder_framework.py:194-250 -> lss.py / lidarnet.py /
// thinktwice_decoder.py:419-489); so does this package's host-side mirror.  A PLAN is that sequence frozen into data: the
// ordered list of C-ABI calls of one forward (entry name + arguments) with every pointer expressed as (buffer, byte offset)
// -- buffer 0 the packed weights, buffer 1 the activation arena, buffers 2.. the inputs -- plus the stream each call runs on
// and the cross-stream dependencies.  The Python mirror is the plan COMPILER (thinktwice_amd/plan.py records one forward);
// this file is the RUNTIME: bind the buffers a host owns, then tt_plan_run() issues the whole half of the forward from C++ on
// caller streams -- no Python, no torch.  Plans serialise (tt_plan_save / tt_plan_load), so a C / C++ / Go / Rust host can
// run the path from a plan file + a weights file (tools/plan_host.cpp does).
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "tt_common.h"

extern "C" int tt_fill_u32(void* dst, long long n_words, unsigned pattern, void* stream) {
    TT_REQUIRE(dst && n_words >= 0, "tt_fill_u32: bad argument");
    if (n_words == 0) return 0;
    if (hipMemsetD32Async((hipDeviceptr_t)dst, (int)pattern, (size_t)n_words, (hipStream_t)stream) != hipSuccess) {
        tt::set_error("tt_fill_u32: hipMemsetD32Async failed");
        return -2;
    }
    return 0;
}

extern "C" int tt_copy_bytes(void* dst, const void* src, long long nbytes, void* stream) {
    TT_REQUIRE(dst && src && nbytes >= 0, "tt_copy_bytes: bad argument");
    if (nbytes == 0) return 0;
    if (hipMemcpyAsync(dst, src, (size_t)nbytes, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) {
        tt::set_error("tt_copy_bytes: hipMemcpyAsync failed");
        return -2;
    }
    return 0;
}

namespace tt {

union PlanVal {
    long long i;
    double d;
    void* p;
};

struct PlanThunk {
    const char* name;
    int (*fn)(const PlanVal*, void*);
    int nargs;
};

#include "plan_thunks.inc"

enum ArgKind : int { kI64 = 0, kF64 = 1, kDevPtr = 2, kBlob = 3, kNull = 4 };

struct PlanArgRec {
    int kind, buffer;          // kDevPtr: buffer id; kBlob: unused
    long long value;           // kI64: value; kDevPtr: byte offset; kBlob: offset into the plan's host blob
    double fvalue;             // kF64
};

struct PlanReloc {             // a device pointer stored INSIDE a host blob (descriptor structs, pointer arrays)
    long long blob_offset;
    int buffer;
    long long offset;
    int op = -1;               // builder only: index of the call whose argument blob holds it (arena liveness; not serialised)
};

struct PlanAlloc {             // builder only: one allocation of the activation arena (tt_plan_add_arena_alloc)
    long long offset, bytes;
};

struct PlanOp {
    int kind;                  // 0 call, 1 sync (stream `a` waits for everything enqueued so far on stream `b`)
    int thunk, stream, a, b;
    int first_arg, nargs;
};

struct PlanBuffer {
    std::string name;
    long long bytes;
};

struct PlanOutput {
    std::string name;
    int buffer;
    long long offset;
    int ndim;
    long long shape[8], stride[8];     // element strides (outputs may be views: a slice of a wider head buffer)
};

}  // namespace tt

struct tt_plan {
    std::vector<tt::PlanOp> ops;
    std::vector<tt::PlanArgRec> args;
    std::vector<tt::PlanReloc> relocs;
    int pending_relocs = 0;      // relocations declared since the last op was added: they belong to the NEXT call
    std::vector<unsigned char> blob;          // pristine host blob (relocations unresolved)
    std::vector<tt::PlanBuffer> buffers;
    std::vector<tt::PlanOutput> outputs;
    std::vector<tt::PlanAlloc> allocs;        // builder only (consumed by tt_plan_compact_arena)
    int nstreams = 1;
    // bound state
    std::vector<unsigned char> bound_blob;
    std::vector<tt::PlanVal> bound_args;
    std::vector<void*> bases;
    std::vector<hipEvent_t> events;
    bool bound = false;
};

using namespace tt;

static int find_thunk(const char* name) {
    const int n = (int)(sizeof(kPlanThunks) / sizeof(kPlanThunks[0]));
    for (int i = 0; i < n; ++i)
        if (strcmp(kPlanThunks[i].name, name) == 0) return i;
    return -1;
}

// Structural checks of a whole plan: everything tt_plan_bind / tt_plan_run index with.  The builder entries enforce these call
// by call; a plan that arrives as a FILE (tt_plan_load: a stale or corrupt plan next to a newer library, tools/plan_host.cpp)
// has to pass them as a whole before it may be bound -- a bad index there would be a write outside a host vector
// (relocations are memcpy'd into the blob) or a read of streams[] / bases[] out of range.
static const int kMaxPlanStreams = 64;
static const char* plan_defect(const tt_plan* p) {
    static thread_local char why[160];
    auto say = [&](const char* fmt, long long a, long long b) { snprintf(why, sizeof(why), fmt, a, b); return (const char*)why; };
    if (p->nstreams < 1 || p->nstreams > kMaxPlanStreams) return say("stream count %lld outside 1..%lld", p->nstreams, kMaxPlanStreams);
    const long long nbuf = (long long)p->buffers.size(), nblob = (long long)p->blob.size(), nargs = (long long)p->args.size();
    for (const PlanBuffer& b : p->buffers)
        if (b.bytes < 0) return say("buffer with a negative size (%lld)%.0lld", b.bytes, 0);
    auto devptr_ok = [&](int buffer, long long offset) {
        return buffer >= 0 && buffer < nbuf && offset >= 0 && offset <= p->buffers[buffer].bytes;
    };
    for (size_t i = 0; i < p->relocs.size(); ++i) {
        const PlanReloc& r = p->relocs[i];
        if (r.blob_offset < 0 || r.blob_offset + 8 > nblob) return say("relocation %lld writes outside the blob (offset %lld)", (long long)i, r.blob_offset);
        if (!devptr_ok(r.buffer, r.offset)) return say("relocation %lld points outside buffer %lld", (long long)i, r.buffer);
    }
    for (size_t i = 0; i < p->ops.size(); ++i) {
        const PlanOp& o = p->ops[i];
        if (o.kind == 1) {
            if (o.a < 0 || o.a >= p->nstreams || o.b < 0 || o.b >= p->nstreams) return say("op %lld: stream dependency on a stream outside 0..%lld", (long long)i, p->nstreams - 1);
            continue;
        }
        if (o.kind != 0) return say("op %lld: unknown kind %lld", (long long)i, o.kind);
        if (o.thunk < 0 || o.thunk >= (int)(sizeof(kPlanThunks) / sizeof(kPlanThunks[0]))) return say("op %lld: entry index %lld", (long long)i, o.thunk);
        if (o.stream < 0 || o.stream >= p->nstreams) return say("op %lld runs on stream %lld, outside the plan's streams", (long long)i, o.stream);
        if (o.nargs != kPlanThunks[o.thunk].nargs || o.first_arg < 0 || (long long)o.first_arg + o.nargs > nargs)
            return say("op %lld: argument range (%lld arguments)", (long long)i, o.nargs);
        for (int k = 0; k < o.nargs; ++k) {
            const PlanArgRec& a = p->args[o.first_arg + k];
            switch (a.kind) {
                case kI64: case kF64: case kNull: break;
                case kBlob:
                    if (a.value < 0 || a.value >= nblob) return say("op %lld: blob argument at offset %lld outside the blob", (long long)i, a.value);
                    break;
                case kDevPtr:
                    if (!devptr_ok(a.buffer, a.value)) return say("op %lld: device pointer outside buffer %lld", (long long)i, a.buffer);
                    break;
                default: return say("op %lld: argument kind %lld", (long long)i, a.kind);
            }
        }
    }
    for (const PlanOutput& o : p->outputs) {
        if (o.ndim < 0 || o.ndim > 8) return say("output with %lld dimensions%.0lld", o.ndim, 0);
        if (o.name == "__decoder_first_op") {
            if (o.offset < 0 || o.offset > (long long)p->ops.size()) return say("decoder marker at op %lld of %lld", o.offset, (long long)p->ops.size());
        } else if (o.buffer >= nbuf || (o.buffer >= 0 && (o.offset < 0 || o.offset > p->buffers[o.buffer].bytes))) {
            return say("output outside buffer %lld (offset %lld)", o.buffer, o.offset);
        }
    }
    return nullptr;
}

extern "C" tt_plan* tt_plan_create(void) { return new tt_plan(); }

extern "C" void tt_plan_destroy(tt_plan* p) {
    if (!p) return;
    for (hipEvent_t e : p->events) (void)hipEventDestroy(e);
    delete p;
}

extern "C" int tt_plan_set_buffer(tt_plan* p, int id, long long bytes, const char* name) {
    TT_REQUIRE(p && id >= 0 && id < 64 && bytes >= 0 && name, "tt_plan_set_buffer: bad argument");
    if ((int)p->buffers.size() <= id) p->buffers.resize(id + 1);
    p->buffers[id].name = name;
    p->buffers[id].bytes = bytes;
    p->bound = false;
    return 0;
}

extern "C" long long tt_plan_add_blob(tt_plan* p, const void* bytes, long long n) {
    if (!p || !bytes || n < 0) return -1;
    const long long off = ((long long)p->blob.size() + 15) / 16 * 16;      // 16 B aligned (descriptor structs)
    p->blob.resize(off + n);
    memcpy(p->blob.data() + off, bytes, (size_t)n);
    p->bound = false;
    return off;
}

extern "C" int tt_plan_add_reloc(tt_plan* p, long long blob_offset, int buffer, long long offset) {
    TT_REQUIRE(p && blob_offset >= 0 && blob_offset + 8 <= (long long)p->blob.size(), "tt_plan_add_reloc: outside the blob");
    // (relocations are declared while the arguments of the NEXT call are assembled: that call's index is ops.size())
    p->relocs.push_back(PlanReloc{blob_offset, buffer, offset, (int)p->ops.size()});
    ++p->pending_relocs;
    p->bound = false;
    return 0;
}

// ---- arena compaction.  The plan compiler serves every temporary of the recorded forward from a bump arena (no reuse: ~6 GB per
// full-size frame), which is what the recorded pointers refer to.  Replayed from C that arena is cold memory on every tick -- each
// temporary a first touch of its pages -- where torch's caching allocator hands the eager forward the same hot blocks again and
// again (round 3: 23.5 ms from the plan against 21.5 eager).  The compiler therefore also declares the allocations
// (tt_plan_add_arena_alloc), and tt_plan_compact_arena re-places them by LIVENESS: an allocation is live from the first to the
// last op that names an address inside it (call arguments and pointers inside argument blobs); outputs stay live to the end.
// Reuse is by stream order only: a block is handed on only between allocations whose EVERY use sits on one and the same
// stream (ops of one stream run in plan order); anything touched from two streams keeps its own memory.
extern "C" int tt_plan_add_arena_alloc(tt_plan* p, long long offset, long long nbytes) {
    TT_REQUIRE(p && offset >= 0 && nbytes >= 0, "tt_plan_add_arena_alloc: bad argument");
    TT_REQUIRE(p->allocs.empty() || offset >= p->allocs.back().offset + p->allocs.back().bytes,
               "tt_plan_add_arena_alloc: allocations must be declared in address order without overlap");
    p->allocs.push_back(PlanAlloc{offset, nbytes});
    return 0;
}

extern "C" long long tt_plan_compact_arena(tt_plan* p, int arena_buffer, long long align) {
    if (!p || arena_buffer < 0 || arena_buffer >= (int)p->buffers.size() || align <= 0) {
        set_error("tt_plan_compact_arena: bad argument");
        return -1;
    }
    const size_t n = p->allocs.size();
    struct Info {
        int first = 0x7fffffff, last = -1, stream = -2;      // stream: -2 unused, -1 several, >= 0 the one stream of every use
        bool pinned = false;
        long long new_off = 0, size = 0;
    };
    std::vector<Info> info(n);
    auto find = [&](long long off) -> long long {            // allocation holding arena offset `off`, or -1
        size_t lo = 0, hi = n;
        while (lo < hi) {
            const size_t mid = (lo + hi) / 2;
            if (p->allocs[mid].offset <= off) lo = mid + 1; else hi = mid;
        }
        if (lo == 0) return -1;
        const PlanAlloc& a = p->allocs[lo - 1];
        return (off < a.offset + (a.bytes > 0 ? a.bytes : 1)) ? (long long)(lo - 1) : -1;
    };
    bool all_found = true;
    auto touch = [&](long long off, int op, int stream) {
        const long long k = find(off);
        if (k < 0) { all_found = false; return; }
        Info& f = info[(size_t)k];
        if (op < f.first) f.first = op;
        if (op > f.last) f.last = op;
        f.stream = f.stream == -2 ? stream : (f.stream == stream ? stream : -1);
    };
    for (size_t i = 0; i < p->ops.size(); ++i) {
        const PlanOp& o = p->ops[i];
        if (o.kind != 0) continue;
        for (int k = 0; k < o.nargs; ++k) {
            const PlanArgRec& a = p->args[o.first_arg + k];
            if (a.kind == kDevPtr && a.buffer == arena_buffer) touch(a.value, (int)i, o.stream);
        }
    }
    for (const PlanReloc& r : p->relocs)
        if (r.buffer == arena_buffer) {
            if (r.op < 0 || r.op >= (int)p->ops.size() || p->ops[r.op].kind != 0) { all_found = false; continue; }
            touch(r.offset, r.op, p->ops[r.op].stream);
        }
    for (const PlanOutput& o : p->outputs)
        if (o.buffer == arena_buffer) {
            const long long k = find(o.offset);
            if (k < 0) all_found = false; else info[(size_t)k].pinned = true;
        }
    if (!all_found) {
        set_error("tt_plan_compact_arena: an arena pointer of the plan lies in no declared allocation (nothing was changed)");
        return -1;
    }
    auto up = [&](long long v) { return (v + align - 1) / align * align; };
    std::vector<size_t> order;
    for (size_t i = 0; i < n; ++i) {
        info[i].size = up(p->allocs[i].bytes);
        if (info[i].stream != -2 || info[i].pinned) order.push_back(i);      // (never-referenced allocations get no memory)
    }
    std::sort(order.begin(), order.end(), [&](size_t a, size_t b) {
        return info[a].first != info[b].first ? info[a].first < info[b].first : a < b;
    });
    struct Block { long long off, size; int stream, free_after; };
    std::vector<Block> free_blocks;
    std::vector<size_t> active;                               // placed, single-stream, not pinned: released once their last op is past
    long long top = 0;
    for (size_t idx : order) {
        Info& f = info[idx];
        for (size_t k = 0; k < active.size();) {              // release what died before this allocation's first use
            const Info& g = info[active[k]];
            if (g.last < f.first) {
                free_blocks.push_back(Block{g.new_off, g.size, g.stream, g.last});
                active[k] = active.back();
                active.pop_back();
            } else {
                ++k;
            }
        }
        long long place = -1;
        if (f.stream >= 0 && !f.pinned && f.size > 0) {
            int best = -1;
            for (size_t k = 0; k < free_blocks.size(); ++k) {
                const Block& b = free_blocks[k];
                if (b.stream == f.stream && b.free_after < f.first && b.size >= f.size &&
                    (best < 0 || b.size < free_blocks[best].size)) best = (int)k;
            }
            if (best >= 0) {
                Block b = free_blocks[best];
                place = b.off;
                if (b.size > f.size) free_blocks[best] = Block{b.off + f.size, b.size - f.size, b.stream, b.free_after};
                else { free_blocks[best] = free_blocks.back(); free_blocks.pop_back(); }
            }
        }
        if (place < 0) { place = top; top += f.size; }
        f.new_off = place;
        if (f.stream >= 0 && !f.pinned && f.size > 0) active.push_back(idx);
    }
    // rewrite every arena pointer
    auto moved = [&](long long off) {
        const long long k = find(off);
        return info[(size_t)k].new_off + (off - p->allocs[(size_t)k].offset);
    };
    for (const PlanOp& o : p->ops) {
        if (o.kind != 0) continue;
        for (int k = 0; k < o.nargs; ++k) {
            PlanArgRec& a = p->args[o.first_arg + k];
            if (a.kind == kDevPtr && a.buffer == arena_buffer) a.value = moved(a.value);
        }
    }
    for (PlanReloc& r : p->relocs)
        if (r.buffer == arena_buffer) r.offset = moved(r.offset);
    for (PlanOutput& o : p->outputs)
        if (o.buffer == arena_buffer) o.offset = moved(o.offset);
    p->allocs.clear();
    p->buffers[arena_buffer].bytes = up(top > 0 ? top : align);
    p->bound = false;
    return p->buffers[arena_buffer].bytes;
}

// kinds[i] / buffers[i] / ivals[i] / fvals[i]: see ArgKind.  The entry's trailing `stream` argument is not listed: the call
// runs on stream slot `stream`.
extern "C" int tt_plan_add_call(tt_plan* p, const char* entry, int nargs, const int* kinds, const int* buffers,
                                const long long* ivals, const double* fvals, int stream) {
    TT_REQUIRE(p && entry && nargs >= 0 && stream >= 0 && stream < kMaxPlanStreams, "tt_plan_add_call: bad argument");
    const int th = find_thunk(entry);
    TT_REQUIRE(th >= 0, "tt_plan_add_call: %s is not a stream-taking entry of this library", entry);
    TT_REQUIRE(kPlanThunks[th].nargs == nargs, "tt_plan_add_call: %s takes %d arguments before its stream, got %d", entry,
               kPlanThunks[th].nargs, nargs);
    PlanOp op{0, th, stream, 0, 0, (int)p->args.size(), nargs};
    for (int i = 0; i < nargs; ++i) p->args.push_back(PlanArgRec{kinds[i], buffers[i], ivals[i], fvals[i]});
    p->ops.push_back(op);
    p->pending_relocs = 0;       // (they were recorded against this op's index)
    if (stream + 1 > p->nstreams) p->nstreams = stream + 1;
    p->bound = false;
    return 0;
}

extern "C" int tt_plan_add_sync(tt_plan* p, int waiter_stream, int signal_stream) {
    TT_REQUIRE(p && waiter_stream >= 0 && signal_stream >= 0 && waiter_stream < kMaxPlanStreams && signal_stream < kMaxPlanStreams,
               "tt_plan_add_sync: bad argument");
    // the liveness analysis of tt_plan_compact_arena attributes a blob's pointers to the op index current when they were
    // declared: a sync slipped in between a call's relocations and the call itself would silently mis-attribute them
    TT_REQUIRE(p->pending_relocs == 0, "tt_plan_add_sync: %d relocation(s) were declared for a call that has not been added yet",
               p->pending_relocs);
    p->ops.push_back(PlanOp{1, -1, 0, waiter_stream, signal_stream, 0, 0});
    const int m = (waiter_stream > signal_stream ? waiter_stream : signal_stream) + 1;
    if (m > p->nstreams) p->nstreams = m;
    p->bound = false;
    return 0;
}

extern "C" int tt_plan_add_output(tt_plan* p, const char* name, int buffer, long long offset, int ndim, const long long* shape,
                                  const long long* stride) {
    TT_REQUIRE(p && name && ndim >= 0 && ndim <= 8, "tt_plan_add_output: bad argument");
    PlanOutput o;
    o.name = name; o.buffer = buffer; o.offset = offset; o.ndim = ndim;
    for (int i = 0; i < 8; ++i) o.shape[i] = (i < ndim && shape) ? shape[i] : 0;
    for (int i = 0; i < 8; ++i) o.stride[i] = (i < ndim && stride) ? stride[i] : 0;
    p->outputs.push_back(o);
    return 0;
}

extern "C" int tt_plan_num_ops(const tt_plan* p) { return p ? (int)p->ops.size() : -1; }
extern "C" int tt_plan_num_calls(const tt_plan* p) {
    if (!p) return -1;
    int n = 0;
    for (const PlanOp& o : p->ops) n += o.kind == 0;
    return n;
}
extern "C" int tt_plan_num_streams(const tt_plan* p) { return p ? p->nstreams : -1; }
extern "C" int tt_plan_num_buffers(const tt_plan* p) { return p ? (int)p->buffers.size() : -1; }
extern "C" long long tt_plan_buffer_bytes(const tt_plan* p, int id) {
    return (p && id >= 0 && id < (int)p->buffers.size()) ? p->buffers[id].bytes : -1;
}
extern "C" const char* tt_plan_buffer_name(const tt_plan* p, int id) {
    return (p && id >= 0 && id < (int)p->buffers.size()) ? p->buffers[id].name.c_str() : nullptr;
}
extern "C" int tt_plan_num_outputs(const tt_plan* p) { return p ? (int)p->outputs.size() : -1; }
extern "C" int tt_plan_output(const tt_plan* p, int i, const char** name, int* buffer, long long* offset, int* ndim,
                              long long* shape8, long long* stride8) {
    TT_REQUIRE(p && i >= 0 && i < (int)p->outputs.size(), "tt_plan_output: index");
    const PlanOutput& o = p->outputs[i];
    if (name) *name = o.name.c_str();
    if (buffer) *buffer = o.buffer;
    if (offset) *offset = o.offset;
    if (ndim) *ndim = o.ndim;
    if (shape8) memcpy(shape8, o.shape, sizeof(o.shape));
    if (stride8) memcpy(stride8, o.stride, sizeof(o.stride));
    return 0;
}

// Resolve every (buffer, offset) against the base addresses the host owns.  bases[id] must hold tt_plan_buffer_bytes(id).
extern "C" int tt_plan_bind(tt_plan* p, void* const* bases, int nbases) {
    TT_REQUIRE(p && bases && nbases >= (int)p->buffers.size(), "tt_plan_bind: %d buffers needed", p ? (int)p->buffers.size() : 0);
    if (const char* why = plan_defect(p)) {
        set_error("tt_plan_bind: malformed plan: %s", why);
        return -1;
    }
    p->bases.assign(bases, bases + nbases);
    p->bound_blob = p->blob;
    auto resolve = [&](int buffer, long long offset, void** out) -> int {
        TT_REQUIRE(buffer >= 0 && buffer < (int)p->bases.size() && p->bases[buffer], "tt_plan_bind: buffer %d is not bound", buffer);
        TT_REQUIRE(offset >= 0 && offset <= p->buffers[buffer].bytes, "tt_plan_bind: offset outside buffer %d", buffer);
        *out = (char*)p->bases[buffer] + offset;
        return 0;
    };
    for (const PlanReloc& r : p->relocs) {
        void* v;
        if (int rc = resolve(r.buffer, r.offset, &v)) return rc;
        memcpy(p->bound_blob.data() + r.blob_offset, &v, sizeof(v));
    }
    p->bound_args.resize(p->args.size());
    for (size_t i = 0; i < p->args.size(); ++i) {
        const PlanArgRec& a = p->args[i];
        PlanVal v;
        v.i = 0;
        switch (a.kind) {
            case kI64: v.i = a.value; break;
            case kF64: v.d = a.fvalue; break;
            case kNull: v.p = nullptr; break;
            case kBlob:
                TT_REQUIRE(a.value >= 0 && a.value < (long long)p->bound_blob.size(), "tt_plan_bind: blob offset");
                v.p = p->bound_blob.data() + a.value;
                break;
            case kDevPtr:
                if (int rc = resolve(a.buffer, a.value, &v.p)) return rc;
                break;
            default: TT_REQUIRE(false, "tt_plan_bind: argument kind %d", a.kind);
        }
        p->bound_args[i] = v;
    }
    while ((int)p->events.size() < 16) {
        hipEvent_t e;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
            set_error("tt_plan_bind: hipEventCreate failed");
            return -2;
        }
        p->events.push_back(e);
    }
    p->bound = true;
    return 0;
}

// Issue ops [first_op, first_op + num_ops) (num_ops < 0: to the end) on the caller's streams.  Asynchronous.
extern "C" int tt_plan_run_range(tt_plan* p, void* const* streams, int nstreams, int first_op, int num_ops) {
    TT_REQUIRE(p && p->bound, "tt_plan_run: bind the buffers first (tt_plan_bind)");
    TT_REQUIRE(streams && nstreams >= p->nstreams, "tt_plan_run: the plan uses %d streams", p->nstreams);
    const int end = num_ops < 0 ? (int)p->ops.size() : first_op + num_ops;
    TT_REQUIRE(first_op >= 0 && end <= (int)p->ops.size(), "tt_plan_run: op range");
    if (int rc = refuse_after_fault("tt_plan_run")) return rc;     // a barrier time-out of an earlier forward is sticky
    int ev = 0;
    for (int i = first_op; i < end; ++i) {
        const PlanOp& o = p->ops[i];
        if (o.kind == 1) {
            hipEvent_t e = p->events[ev++ % (int)p->events.size()];
            if (hipEventRecord(e, (hipStream_t)streams[o.b]) != hipSuccess ||
                hipStreamWaitEvent((hipStream_t)streams[o.a], e, 0) != hipSuccess) {
                set_error("tt_plan_run: stream dependency %d <- %d failed", o.a, o.b);
                return -2;
            }
            continue;
        }
        const int rc = kPlanThunks[o.thunk].fn(p->bound_args.data() + o.first_arg, streams[o.stream]);
        if (rc != 0) return rc;          // (the entry has set the error text)
    }
    return 0;
}

extern "C" int tt_plan_run(tt_plan* p, void* const* streams, int nstreams) {
    return tt_plan_run_range(p, streams, nstreams, 0, -1);
}

// SURVEY 8b B3's two composites: the encoder half (camera + LiDAR encoders, BEV fusion + flatten:
// encoder_decoder_framework.py:194-235) and the decoder half (coarse heads + five look-and-refine stages:
// thinktwice_decoder.py:419-489) of a bound forward plan, as the plan compiler marked them (output "__decoder_first_op").
static int decoder_first_op(const tt_plan* p) {
    for (const PlanOutput& o : p->outputs)
        if (o.name == "__decoder_first_op") return (int)o.offset;
    return -1;
}

extern "C" int tt_encoder_fwd(tt_plan* p, void* const* streams, int nstreams) {
    TT_REQUIRE(p, "tt_encoder_fwd: null plan");
    const int d = decoder_first_op(p);
    TT_REQUIRE(d >= 0, "tt_encoder_fwd: the plan does not mark its decoder half");
    return tt_plan_run_range(p, streams, nstreams, 0, d);
}

extern "C" int tt_decoder_fwd(tt_plan* p, void* const* streams, int nstreams) {
    TT_REQUIRE(p, "tt_decoder_fwd: null plan");
    const int d = decoder_first_op(p);
    TT_REQUIRE(d >= 0, "tt_decoder_fwd: the plan does not mark its decoder half");
    return tt_plan_run_range(p, streams, nstreams, d, -1);
}

// ---- serialisation: little-endian, versioned; entries by NAME (a plan survives library rebuilds that keep the ABI)
namespace {
struct Writer {
    FILE* f;
    bool ok = true;
    void raw(const void* p, size_t n) { ok = ok && fwrite(p, 1, n, f) == n; }
    void i64(long long v) { raw(&v, 8); }
    void f64(double v) { raw(&v, 8); }
    void str(const std::string& s) { i64((long long)s.size()); raw(s.data(), s.size()); }
};
struct Reader {
    FILE* f;
    bool ok = true;
    void raw(void* p, size_t n) { ok = ok && fread(p, 1, n, f) == n; }
    long long i64() { long long v = 0; raw(&v, 8); return v; }
    double f64() { double v = 0; raw(&v, 8); return v; }
    std::string str() {
        const long long n = i64();
        std::string s;
        if (ok && n >= 0 && n < (1 << 20)) { s.resize((size_t)n); raw(&s[0], (size_t)n); } else ok = false;
        return s;
    }
};
const long long kMagic = 0x314E414C50545454ll;   // "TTTPLAN1"
}  // namespace

extern "C" int tt_plan_save(const tt_plan* p, const char* path) {
    TT_REQUIRE(p && path, "tt_plan_save: null");
    FILE* f = fopen(path, "wb");
    TT_REQUIRE(f, "tt_plan_save: cannot open %s", path);
    Writer w{f};
    w.i64(kMagic);
    w.i64(p->nstreams);
    w.i64((long long)p->buffers.size());
    for (const PlanBuffer& b : p->buffers) { w.str(b.name); w.i64(b.bytes); }
    w.i64((long long)p->blob.size());
    w.raw(p->blob.data(), p->blob.size());
    w.i64((long long)p->relocs.size());
    for (const PlanReloc& r : p->relocs) { w.i64(r.blob_offset); w.i64(r.buffer); w.i64(r.offset); }
    w.i64((long long)p->ops.size());
    for (const PlanOp& o : p->ops) {
        w.i64(o.kind);
        if (o.kind == 1) { w.i64(o.a); w.i64(o.b); continue; }
        w.str(kPlanThunks[o.thunk].name);
        w.i64(o.stream);
        w.i64(o.nargs);
        for (int i = 0; i < o.nargs; ++i) {
            const PlanArgRec& a = p->args[o.first_arg + i];
            w.i64(a.kind); w.i64(a.buffer); w.i64(a.value); w.f64(a.fvalue);
        }
    }
    w.i64((long long)p->outputs.size());
    for (const PlanOutput& o : p->outputs) {
        w.str(o.name); w.i64(o.buffer); w.i64(o.offset); w.i64(o.ndim);
        for (int i = 0; i < 8; ++i) w.i64(o.shape[i]);
        for (int i = 0; i < 8; ++i) w.i64(o.stride[i]);
    }
    const bool ok = w.ok;
    fclose(f);
    TT_REQUIRE(ok, "tt_plan_save: short write to %s", path);
    return 0;
}

extern "C" tt_plan* tt_plan_load(const char* path) {
    FILE* f = path ? fopen(path, "rb") : nullptr;
    if (!f) { set_error("tt_plan_load: cannot open %s", path ? path : "(null)"); return nullptr; }
    Reader r{f};
    tt_plan* p = new tt_plan();
    auto fail = [&](const char* why) { set_error("tt_plan_load: %s (%s)", why, path); fclose(f); delete p; return (tt_plan*)nullptr; };
    if (r.i64() != kMagic) return fail("not a plan file");
    p->nstreams = (int)r.i64();
    const long long nb = r.i64();
    if (!r.ok || nb < 0 || nb > 64) return fail("buffer table");
    for (long long i = 0; i < nb; ++i) { PlanBuffer b; b.name = r.str(); b.bytes = r.i64(); p->buffers.push_back(b); }
    const long long blob = r.i64();
    if (!r.ok || blob < 0 || blob > (1ll << 32)) return fail("blob size");
    p->blob.resize((size_t)blob);
    r.raw(p->blob.data(), (size_t)blob);
    const long long nr = r.i64();
    if (!r.ok || nr < 0 || nr > (1ll << 28)) return fail("relocations");
    for (long long i = 0; i < nr; ++i) { PlanReloc x; x.blob_offset = r.i64(); x.buffer = (int)r.i64(); x.offset = r.i64(); p->relocs.push_back(x); }
    const long long no = r.i64();
    if (!r.ok || no < 0 || no > (1ll << 24)) return fail("op count");
    for (long long i = 0; i < no; ++i) {
        PlanOp o{};
        o.kind = (int)r.i64();
        if (o.kind == 1) { o.a = (int)r.i64(); o.b = (int)r.i64(); p->ops.push_back(o); continue; }
        const std::string name = r.str();
        o.thunk = find_thunk(name.c_str());
        if (o.thunk < 0) return fail("unknown entry in the plan (library / plan version mismatch)");
        o.stream = (int)r.i64();
        o.nargs = (int)r.i64();
        if (!r.ok || o.nargs != kPlanThunks[o.thunk].nargs) return fail("argument count mismatch");
        o.first_arg = (int)p->args.size();
        for (int k = 0; k < o.nargs; ++k) {
            PlanArgRec a;
            a.kind = (int)r.i64(); a.buffer = (int)r.i64(); a.value = r.i64(); a.fvalue = r.f64();
            p->args.push_back(a);
        }
        p->ops.push_back(o);
    }
    const long long nout = r.i64();
    if (!r.ok || nout < 0 || nout > 4096) return fail("outputs");
    for (long long i = 0; i < nout; ++i) {
        PlanOutput o;
        o.name = r.str(); o.buffer = (int)r.i64(); o.offset = r.i64(); o.ndim = (int)r.i64();
        for (int k = 0; k < 8; ++k) o.shape[k] = r.i64();
        for (int k = 0; k < 8; ++k) o.stride[k] = r.i64();
        p->outputs.push_back(o);
    }
    if (!r.ok) return fail("truncated file");
    if (const char* why = plan_defect(p)) return fail(why);
    fclose(f);
    return p;
}
