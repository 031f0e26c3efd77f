// comm.cpp — the one exchange step of the path behind the C ABI: a row-sharded gallery's top-k over RCCL.
//
// What it stands for in the reference: FaceRecognizer::compareFaces (src/face_recognizer.cpp:320-334) called in a loop over the
// enrolled features (src/main.cpp:221-238), with the enrolled set split across the GPUs of one node.  Everything else on the path is
// per-frame independent and needs no collective.  Per call and rank: all-gather of the query rows (world * nq * dim floats: 128 KB at
// 64 queries), the local scan (gallery.hip), ONE all-gather of the per-rank (score | index) planes (2 * Q * k words: 8 KB per rank at
// Q = 64, k = 16 — latency-bound over xGMI, so both planes travel together), then the merge kernel the single-gallery call ends with.
// All of it is queued on the caller's stream; the host never touches the data.
//
// librccl is loaded with dlopen on first use: libfacehip.so itself links the HIP runtime only, single-GPU callers never load RCCL,
// and a process that already carries RCCL (torch.distributed) shares that copy (same soname).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>

#include "../../include/facehip.h"
#include "engine.h"

namespace fh {
void set_error(const std::string& msg);

namespace {
struct Rccl {
    void* so = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string why;
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.so = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.so) break;
        }
        if (!r.so) { r.why = std::string("librccl not loadable: ") + (dlerror() ? dlerror() : "?"); return; }
        auto sym = [&](const char* n) { void* p = dlsym(r.so, n); if (!p && r.why.empty()) r.why = std::string("librccl lacks ") + n; return p; };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    });
    if (!r.why.empty()) throw std::runtime_error("comm: " + r.why);
    return r;
}

void nccl_check(ncclResult_t rc, const char* what) {
    if (rc == ncclSuccess) return;
    Rccl& r = rccl();
    throw std::runtime_error(std::string("comm: ") + what + " failed: " + (r.GetErrorString ? r.GetErrorString(rc) : "rccl error"));
}
}  // namespace
}  // namespace fh

namespace fh { Gallery& gallery_of(fh_gallery* g); }

static_assert(FH_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "fh_comm id = ncclUniqueId");

struct fh_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    fh::DevBuf qall, planes, gathered;        // gathered queries; local [scores | indices] planes; every rank's planes
    ~fh_comm() {
        if (comm) {
            (void)hipSetDevice(device);
            (void)hipDeviceSynchronize();
            (void)fh::rccl().CommDestroy(comm);
        }
    }
};

namespace {
template <class F>
int comm_guarded(F&& f) {
    try {
        return f();
    } catch (const std::exception& e) {
        fh::set_error(e.what());
        const std::string m = e.what();
        return m.rfind("HIP error", 0) == 0 ? FH_ERR_DEVICE : FH_ERR_STATE;
    } catch (...) {
        fh::set_error("unknown exception");
        return FH_ERR_STATE;
    }
}
int comm_arg(const char* msg) { fh::set_error(msg); return FH_ERR_ARG; }
}  // namespace

extern "C" {

int fh_comm_unique_id(unsigned char id[FH_COMM_ID_BYTES]) {
    if (!id) return comm_arg("fh_comm_unique_id: null id");
    return comm_guarded([&] {
        ncclUniqueId u;
        fh::nccl_check(fh::rccl().GetUniqueId(&u), "ncclGetUniqueId");
        std::memcpy(id, u.internal, FH_COMM_ID_BYTES);
        return FH_OK;
    });
}

fh_comm* fh_comm_create(int rank, int world, const unsigned char id[FH_COMM_ID_BYTES], int device) {
    if (!id || world <= 0 || rank < 0 || rank >= world || device < 0) { fh::set_error("fh_comm_create: bad argument"); return nullptr; }
    fh_comm* c = nullptr;
    const int rc = comm_guarded([&] {
        FH_HIP(hipSetDevice(device));
        ncclUniqueId u;
        std::memcpy(u.internal, id, FH_COMM_ID_BYTES);
        c = new fh_comm;
        c->rank = rank; c->world = world; c->device = device;
        fh::nccl_check(fh::rccl().CommInitRank(&c->comm, world, u, rank), "ncclCommInitRank");
        return FH_OK;
    });
    if (rc != FH_OK) { if (c) { c->comm = nullptr; delete c; } return nullptr; }
    return c;
}

void fh_comm_destroy(fh_comm* c) { delete c; }
int fh_comm_rank(const fh_comm* c) { return c ? c->rank : comm_arg("fh_comm_rank: null handle"); }
int fh_comm_world(const fh_comm* c) { return c ? c->world : comm_arg("fh_comm_world: null handle"); }

int fh_comm_allgather_f32_dev(fh_comm* c, const float* d_send, float* d_recv, long long count, void* stream) {
    if (!c || !d_send || !d_recv || count <= 0) return comm_arg("fh_comm_allgather_f32_dev: bad argument");
    return comm_guarded([&] {
        fh::nccl_check(fh::rccl().AllGather(d_send, d_recv, (size_t)count, ncclFloat, c->comm, reinterpret_cast<hipStream_t>(stream)), "ncclAllGather");
        return c->world;
    });
}

int fh_gallery_topk_sharded_dev(fh_gallery* g, fh_comm* c, const float* d_q, int nq, int k, float* d_scores, int* d_indices, void* stream) {
    if (!g || !c || !d_q || !d_scores || !d_indices) return comm_arg("fh_gallery_topk_sharded_dev: null argument");
    if (nq <= 0 || k <= 0 || k > 16 || (long)c->world * k > 65536) return comm_arg("fh_gallery_topk_sharded_dev: bad size");
    return comm_guarded([&] {
        fh::Gallery& gal = fh::gallery_of(g);
        hipStream_t s = reinterpret_cast<hipStream_t>(stream);
        const int W = c->world, Q = W * nq, dim = gal.dim();
        const size_t plane = (size_t)Q * k;
        c->qall.ensure((size_t)Q * dim * sizeof(float));
        c->planes.ensure(2 * plane * sizeof(float));
        c->gathered.ensure((size_t)W * 2 * plane * sizeof(float));
        // (1) every rank's queries to every rank
        fh::nccl_check(fh::rccl().AllGather(d_q, c->qall.p, (size_t)nq * dim, ncclFloat, c->comm, s), "ncclAllGather(queries)");
        // (2) all Q queries against the LOCAL shard, <= 256 per scan; scores into plane 0, global indices into plane 1
        float* ls = c->planes.as<float>();
        int* li = reinterpret_cast<int*>(ls + plane);
        for (int q0 = 0; q0 < Q; q0 += 256) {
            const int n = std::min(256, Q - q0);
            gal.topk_dev(c->qall.as<float>() + (size_t)q0 * dim, n, k, ls + (size_t)q0 * k, li + (size_t)q0 * k, s);
        }
        // (3) ONE collective for both planes (indices travel as raw 32-bit words), rank-major
        fh::nccl_check(fh::rccl().AllGather(c->planes.p, c->gathered.p, 2 * plane, ncclFloat, c->comm, s), "ncclAllGather(top-k)");
        // (4) merge: part w = rank w's planes, 2 * plane words apart
        const float* gs = c->gathered.as<float>();
        fh::launch_topk_merge_strided(gs, reinterpret_cast<const int*>(gs + plane), W, Q, k, (long)(2 * plane), d_scores, d_indices, s);
        FH_HIP(hipGetLastError());
        return Q;
    });
}

}  // extern "C"
