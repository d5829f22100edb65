// libnequip_amd_torch.so -- the `torch.ops.nequip_amd.*` dispatcher ops registered from C++ (TORCH_LIBRARY), for
// runtimes that load an AOTInductor package of the HIP-backed model WITHOUT a Python interpreter (a LAMMPS pair style,
// a C++ server).
//
// The reference ships its accelerated kernels to such runtimes the same way: the compiled `.nequip.pt2` package names the
// libraries that define its custom ops (`nequip_custom_ops_libs`, nequip/utils/aoti_metadata.py:24-54; the OpenEquivariance
// adapter registers its TorchScript-free ops from a shared object, nequip/nn/_tp_scatter_oeq.py:13-47) and the loader
// makes them known to the dispatcher before it runs the package (nequip/model/inference_models/aotinductor.py:57-125).
//
// Scope: the ops an exported energy / forces / virial evaluation contains -- first derivatives are explicit nodes of such
// a graph, so no autograd formulas are needed here: edge_vectors(+_adj), edge_embed_fwd/_bwd, radial_mlp_fwd/_bwd,
// tp_scatter_fwd/_bwd, node_linear, gate, gate_bwd.  Same schemas and argument meaning as the Python registrations
// (nequip_amd/nn/_tp_scatter_ops.py, _mlp_ops.py, _edge_vector_ops.py, embedding/_edge_ops.py, o3/_node_ops.py), same
// C-ABI calls underneath (include/nequip_amd.h); the second-order ops of the training path stay Python-registered.
// A process must load ONE of the two registrations: this library refuses to define the ops twice (it then leaves the
// existing registration alone and says so once on stderr).
//
// PyTorch supplies tensors, the caching allocator and the current stream here; every kernel is libnequip_amd.so's.
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <c10/core/DeviceGuard.h>
#include <c10/util/intrusive_ptr.h>
#include <torch/library.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <atomic>
#include <mutex>
#include <sstream>
#include <string>
#include <tuple>
#include <vector>

#include "nequip_amd.h"
#include "nequip_amd_torch.h"

namespace {

using at::Tensor;
using OptTensor = c10::optional<at::Tensor>;

#define NQA_CALL(expr, what)                                                                              \
  do {                                                                                                    \
    const int rc_ = (expr);                                                                               \
    TORCH_CHECK(rc_ == NQA_OK, "nequip_amd: ", what, " failed (", rc_, "): ", nqa_last_error());          \
  } while (0)

void* ptr(const Tensor& t) { return t.defined() ? t.data_ptr() : nullptr; }
void* ptr(const OptTensor& t) { return (t.has_value() && t->defined()) ? t->data_ptr() : nullptr; }

nqa_stream stream_of(const Tensor& t) {
  return static_cast<nqa_stream>(c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.get_device()).stream());
}

void require_gpu(const Tensor& t, const char* op) {
  TORCH_CHECK(t.is_cuda(), "nequip_amd::", op, ": HIP kernels only (got a ", t.device().str(), " tensor); there is no CPU fallback");
}

int32_t nqa_dtype(const Tensor& t, const char* op) {
  if (t.scalar_type() == at::kFloat) return NQA_F32;
  if (t.scalar_type() == at::kDouble) return NQA_F64;
  TORCH_CHECK(false, "nequip_amd::", op, ": float32 / float64 model dtypes only, got ", t.scalar_type());
}

bool env_on(const char* name) {
  const char* v = std::getenv(name);
  return v != nullptr && v[0] != '\0' && !(v[0] == '0' && v[1] == '\0');
}

std::vector<std::string> split(const std::string& s, char sep) {
  std::vector<std::string> out;
  std::string cur;
  for (char c : s) {
    if (c == sep) {
      out.push_back(cur);
      cur.clear();
    } else {
      cur.push_back(c);
    }
  }
  out.push_back(cur);
  return out;
}

// ---- irreps text ("64x0e+64x1o", as nequip_amd.o3.Irreps prints it) -------------------------------------------------------
struct Ir {
  int32_t mul, l, p;
  int32_t d() const { return 2 * l + 1; }
  int64_t dim() const { return (int64_t)mul * d(); }
};

std::vector<Ir> parse_irreps(const std::string& text) {
  std::vector<Ir> out;
  for (const std::string& rec0 : split(text, '+')) {
    std::string rec;
    for (char c : rec0)
      if (c != ' ') rec.push_back(c);
    if (rec.empty()) continue;
    const size_t x = rec.find('x');
    TORCH_CHECK(x != std::string::npos && rec.size() >= x + 3, "nequip_amd: cannot parse irrep '", rec, "'");
    const char pc = rec.back();
    TORCH_CHECK(pc == 'e' || pc == 'o', "nequip_amd: cannot parse irrep '", rec, "'");
    Ir ir;
    ir.mul = std::stoi(rec.substr(0, x));
    ir.l = std::stoi(rec.substr(x + 1, rec.size() - x - 2));
    ir.p = pc == 'e' ? 1 : -1;
    out.push_back(ir);
  }
  return out;
}

int64_t irreps_dim(const std::vector<Ir>& v) {
  int64_t n = 0;
  for (const Ir& ir : v) n += ir.dim();
  return n;
}

std::vector<int32_t> irreps_offsets(const std::vector<Ir>& v) {
  std::vector<int32_t> off(v.size() + 1, 0);
  for (size_t i = 0; i < v.size(); ++i) off[i + 1] = off[i] + (int32_t)v[i].dim();
  return off;
}

std::mutex& registry_mutex() {
  static std::mutex m;
  return m;
}

// ---- derived constants of weight tensors, keyed on the identity of the weight's storage (nequip_amd/utils/constcache.py) -----
// An AOTInductor package hands its constant buffers to these ops at every call: packed / transposed / split images of a
// weight are built once per (storage -- held by the entry, so its address cannot be recycled under it --, data pointer,
// version counter, shape, tag).  A weight tensor that is recomputed per call never hits and only costs a slot of the
// bounded list.  NQA_OP_CONSTANT_CACHE=0 switches the cache off.
struct ConstEntry {
  const void* storage;
  const void* data;
  uint32_t version;
  int64_t rows, cols;
  std::string tag;
  c10::Storage keep;
  Tensor value;
};

std::pair<Tensor, bool> const_cached(const Tensor& t, const std::string& tag, const std::function<Tensor()>& build) {
  static std::vector<ConstEntry> entries;  // most recent last; guarded by registry_mutex()
  static const bool enabled = [] {
    const char* v = std::getenv("NQA_OP_CONSTANT_CACHE");
    return !(v != nullptr && v[0] == '0');
  }();
  if (!enabled) return {build(), false};
  const void* st = t.storage().unsafeGetStorageImpl();
  const int64_t rows = t.dim() > 0 ? t.size(0) : 1, cols = t.numel();
  {
    std::lock_guard<std::mutex> lock(registry_mutex());
    for (size_t i = entries.size(); i-- > 0;) {
      ConstEntry& e = entries[i];
      if (e.storage == st && e.data == t.data_ptr() && e.version == t._version() && e.rows == rows && e.cols == cols &&
          e.tag == tag) {
        Tensor hit = e.value;
        if (i + 1 != entries.size()) std::rotate(entries.begin() + (long)i, entries.begin() + (long)i + 1, entries.end());
        return {hit, true};
      }
    }
  }
  Tensor value = build();  // (outside the lock: builders use the registry themselves)
  std::lock_guard<std::mutex> lock(registry_mutex());
  entries.push_back(ConstEntry{st, t.data_ptr(), t._version(), rows, cols, tag, t.storage(), value});
  if (entries.size() > 512) entries.erase(entries.begin());
  return {value, false};
}

// ---- tensor-product plans: text -> nqa_plan + per-device table image ----------------------------------------------------
struct Plan {
  nqa_plan* handle = nullptr;
  std::vector<uint8_t> image_host;
  std::map<int, Tensor> image;  // by device index
  int64_t dim_in1 = 0, dim_in2 = 0, dim_out = 0, weight_numel = 0;
  bool out_needs_zero = false, prefer_fused_bwd = false, fused_rows_ok = false;
};

Plan& plan_of(const std::string& key) {
  static std::map<std::string, std::unique_ptr<Plan>> plans;
  std::lock_guard<std::mutex> lock(registry_mutex());
  auto it = plans.find(key);
  if (it != plans.end()) return *it->second;
  const auto parts = split(key, '|');
  TORCH_CHECK(parts.size() == 4, "nequip_amd: malformed tensor-product plan text");
  const auto i1 = parse_irreps(parts[0]), i2 = parse_irreps(parts[1]), io = parse_irreps(parts[2]);
  std::vector<int32_t> a1, a2, ao;
  std::vector<double> pw;
  for (const std::string& rec : split(parts[3], ';')) {
    if (rec.empty()) continue;
    const auto f = split(rec, ',');
    TORCH_CHECK(f.size() == 4, "nequip_amd: malformed instruction '", rec, "'");
    a1.push_back(std::stoi(f[0]));
    a2.push_back(std::stoi(f[1]));
    ao.push_back(std::stoi(f[2]));
    pw.push_back(std::stod(f[3]));
  }
  auto cols = [](const std::vector<Ir>& v, std::vector<int32_t>& mul, std::vector<int32_t>& l, std::vector<int32_t>& p) {
    for (const Ir& ir : v) {
      mul.push_back(ir.mul);
      l.push_back(ir.l);
      p.push_back(ir.p);
    }
    if (mul.empty()) {  // (never dereferenced for n = 0; keeps .data() non-null)
      mul.push_back(0), l.push_back(0), p.push_back(1);
    }
  };
  std::vector<int32_t> m1, l1, p1, m2, l2, p2, mo, lo, po;
  cols(i1, m1, l1, p1);
  cols(i2, m2, l2, p2);
  cols(io, mo, lo, po);
  const int32_t n = (int32_t)a1.size();
  if (n == 0) a1.push_back(0), a2.push_back(0), ao.push_back(0), pw.push_back(1.0);
  auto P = std::make_unique<Plan>();
  NQA_CALL(nqa_plan_create((int32_t)i1.size(), m1.data(), l1.data(), p1.data(), (int32_t)i2.size(), m2.data(), l2.data(),
                           p2.data(), (int32_t)io.size(), mo.data(), lo.data(), po.data(), n, a1.data(), a2.data(),
                           ao.data(), pw.data(), NQA_LAYOUT_MUL_IR, NQA_LAYOUT_MUL_IR, &P->handle),
           "nqa_plan_create");
  P->dim_in1 = nqa_plan_query(P->handle, NQA_PLAN_DIM_IN1);
  P->dim_in2 = nqa_plan_query(P->handle, NQA_PLAN_DIM_IN2);
  P->dim_out = nqa_plan_query(P->handle, NQA_PLAN_DIM_OUT);
  P->weight_numel = nqa_plan_query(P->handle, NQA_PLAN_WEIGHT_NUMEL);
  P->out_needs_zero = nqa_plan_query(P->handle, NQA_PLAN_OUT_NEEDS_ZERO) != 0;
  P->fused_rows_ok = nqa_plan_query(P->handle, NQA_PLAN_FUSED_ROWS_OK) != 0;
  // same rule as nequip_amd/nn/_tp_scatter_base.py (_Kernels.prefer_fused_bwd)
  P->prefer_fused_bwd = (P->weight_numel + P->dim_out) >= 3 * P->dim_in1;
  const int64_t nbytes = nqa_plan_image_bytes(P->handle);
  TORCH_CHECK(nbytes > 0, "nequip_amd: nqa_plan_image_bytes failed");
  P->image_host.resize((size_t)nbytes);
  NQA_CALL(nqa_plan_image_write(P->handle, P->image_host.data(), nbytes), "nqa_plan_image_write");
  Plan& ref = *P;
  plans.emplace(key, std::move(P));
  return ref;
}

Tensor bytes_on_device(const std::vector<uint8_t>& host, const at::Device& device) {
  Tensor h = at::empty({(int64_t)std::max<size_t>(host.size(), 1)}, at::TensorOptions().dtype(at::kByte));
  if (!host.empty()) std::memcpy(h.data_ptr(), host.data(), host.size());
  return h.to(device);
}

Tensor plan_image(Plan& P, const at::Device& device) {
  std::lock_guard<std::mutex> lock(registry_mutex());
  auto it = P.image.find(device.index());
  if (it == P.image.end()) it = P.image.emplace(device.index(), bytes_on_device(P.image_host, device)).first;
  return it->second;
}

// ---- edge topology: CSR by destination and by source, shared by the ops of one evaluation ------------------------------
// An entry is keyed on the identity of the STORAGE behind the two index tensors (held through weak references, so an
// address cannot be recycled under an entry), the version counter that views share with their base, data pointers,
// strides and sizes -- the rule of nequip_amd/nn/_topology.py.  Every `edge_index[0]` / `edge_index[1]` view of one input
// tensor maps to the same entry, whether the graph or one of these ops made the view.
//
// Reuse contract (include/nequip_amd_torch.h).  This library serves hosts without an interpreter, and those commonly
// refill ONE persistent device buffer with raw hipMemcpy / Kokkos kernels -- which bumps no version counter.  So the
// DEFAULT (mode 1) reuses an entry only within one evaluation: `edge_embed_fwd` -- run exactly once, at the top of every
// exported energy graph, before any op that needs a CSR -- starts a new evaluation (so does nqa_torch_begin_evaluation()),
// and entries of earlier evaluations are never hit.  Mode 2 (NQA_TOPOLOGY_CACHE=2 / nqa_torch_topology_cache_mode(2)) is
// the opt-in for hosts whose index tensors only ever change through torch: entries live across evaluations as in the
// Python host.  Mode 0 (NQA_TOPOLOGY_CACHE=0): never reuse.  NQA_TOPOLOGY_VERIFY=1 compares a device checksum of the two
// index tensors on every hit (one host synchronisation per hit: diagnosis, not production) and rebuilds on a mismatch.
std::atomic<int>& cache_mode() {
  static std::atomic<int> mode([] {
    const char* v = std::getenv("NQA_TOPOLOGY_CACHE");
    if (v == nullptr || v[0] == '\0') return 1;
    return v[0] == '0' ? 0 : (v[0] == '2' ? 2 : 1);
  }());
  return mode;
}
std::atomic<uint64_t>& evaluation_generation() {
  static std::atomic<uint64_t> g(1);
  return g;
}
bool verify_hits() {
  static const bool v = [] {
    const char* e = std::getenv("NQA_TOPOLOGY_VERIFY");
    return e != nullptr && e[0] != '\0' && e[0] != '0';
  }();
  return v;
}
int64_t index_checksum(const Tensor& dst, const Tensor& src) {
  if (dst.numel() == 0) return 0;
  return (dst * 1000003 + src).sum().item<int64_t>();  // (int64 wrap-around is fine for a checksum)
}
struct Csr {
  Tensor rowptr, edge_id, other;
};
// nequip_amd/nn/_topology.py::EdgePairing: weight rows of a paired list (rows[e] = p for the representative edge of pair p,
// p + P for its reverse), the representative edges, the rows in the slot order of the two CSRs, and the owner lists of the
// pair-centric backward (build_owner_csr)
struct Pairing {
  Tensor rows, rep, partner;
  int64_t P = 0;
  Tensor slots_dst, slots_src;
  Tensor owner[7];
  bool has_owner = false;
};

struct Topology {
  c10::weak_intrusive_ptr<c10::StorageImpl> dst_ref, src_ref;
  const void *dst_ptr, *src_ptr;
  uint32_t dst_version, src_version;
  int64_t num_nodes, num_edges, dst_stride, src_stride;
  uint64_t generation = 0;  // the evaluation that built the entry
  int64_t checksum = 0;     // NQA_TOPOLOGY_VERIFY
  Tensor dst, src;  // contiguous int64
  Csr by_dst, by_src;
  bool has_dst = false, has_src = false;
  bool pairing_done = false;           // reverse-edge pairing (radial_tp_*): verdict read once per entry
  const void* pairing_shift = nullptr;
  uint32_t pairing_shift_version = 0;  // Tensor::_version() of the shift tensor the pairing was decided on
  int64_t pairing_shift_numel = -1;
  Tensor pairing_shift_hold;  // keeps the storage alive: a freed and re-used address cannot alias the key
  std::shared_ptr<struct Pairing> pairing;
  std::mutex build;  // the CSRs are built on first use, outside the registry lock
  Topology(const Tensor& d, const Tensor& s)
      : dst_ref(c10::intrusive_ptr<c10::StorageImpl>::reclaim_copy(d.storage().unsafeGetStorageImpl())),
        src_ref(c10::intrusive_ptr<c10::StorageImpl>::reclaim_copy(s.storage().unsafeGetStorageImpl())) {}
};

Csr build_csr(const Tensor& key, const Tensor& other, int64_t N, int64_t E) {
  Csr c;
  const auto opt = key.options().dtype(at::kInt);
  c.rowptr = at::empty({N + 1}, opt);
  c.edge_id = at::empty({std::max<int64_t>(E, 1)}, opt);
  c.other = at::empty({std::max<int64_t>(E, 1)}, opt);
  const int64_t ws_bytes = nqa_csr_workspace_bytes(N, E);
  TORCH_CHECK(ws_bytes >= 0, "nequip_amd: edge list exceeds the int32 index range supported by the kernels");
  Tensor ws = at::empty({std::max<int64_t>(ws_bytes, 1)}, key.options().dtype(at::kByte));
  NQA_CALL(nqa_csr_build(static_cast<const int64_t*>(key.data_ptr()), static_cast<const int64_t*>(other.data_ptr()), N, E,
                         static_cast<int32_t*>(c.rowptr.data_ptr()), static_cast<int32_t*>(c.edge_id.data_ptr()),
                         static_cast<int32_t*>(c.other.data_ptr()), nullptr, ws.data_ptr(), ws_bytes, stream_of(key)),
           "nqa_csr_build");
  return c;
}

std::vector<std::shared_ptr<Topology>>& topology_cache_entries() {  // most recent last; guarded by registry_mutex()
  static std::vector<std::shared_ptr<Topology>> cache;
  return cache;
}

std::shared_ptr<Topology> topology_of(const Tensor& edge_dst, const Tensor& edge_src, int64_t num_nodes) {
  TORCH_CHECK(edge_dst.scalar_type() == at::kLong && edge_src.scalar_type() == at::kLong, "nequip_amd: edge indices must be int64");
  TORCH_CHECK(edge_dst.dim() == 1 && edge_dst.sizes() == edge_src.sizes(), "nequip_amd: edge_dst / edge_src must be 1-D of one length");
  const int mode = cache_mode().load();
  const bool reuse = mode != 0;
  const uint64_t gen = evaluation_generation().load();
  std::lock_guard<std::mutex> lock(registry_mutex());
  std::vector<std::shared_ptr<Topology>>& cache = topology_cache_entries();
  if (reuse) {
    for (size_t i = 0; i < cache.size(); ++i) {
      Topology& t = *cache[i];
      auto d = t.dst_ref.lock();
      auto s = t.src_ref.lock();
      if (mode == 1 && t.generation != gen) continue;  // built by an earlier evaluation: its buffers may have been refilled
      if (d && s && d.get() == edge_dst.storage().unsafeGetStorageImpl() &&
          s.get() == edge_src.storage().unsafeGetStorageImpl() && t.dst_version == edge_dst._version() &&
          t.src_version == edge_src._version() && t.dst_ptr == edge_dst.data_ptr() && t.src_ptr == edge_src.data_ptr() &&
          t.num_edges == edge_dst.numel() && t.num_nodes == num_nodes &&
          (t.num_edges == 0 || (t.dst_stride == edge_dst.stride(0) && t.src_stride == edge_src.stride(0)))) {
        auto hit = cache[i];
        cache.erase(cache.begin() + (long)i);
        if (verify_hits() && index_checksum(edge_dst, edge_src) != hit->checksum) break;  // rewritten behind torch's back
        cache.push_back(hit);
        return hit;
      }
    }
  }
  auto t = std::make_shared<Topology>(edge_dst, edge_src);
  t->dst_ptr = edge_dst.data_ptr();
  t->src_ptr = edge_src.data_ptr();
  t->dst_version = edge_dst._version();
  t->src_version = edge_src._version();
  t->num_nodes = num_nodes;
  t->num_edges = edge_dst.numel();
  t->dst_stride = edge_dst.stride(0);
  t->src_stride = edge_src.stride(0);
  t->dst = edge_dst.contiguous();
  t->src = edge_src.contiguous();
  t->generation = gen;
  if (verify_hits()) t->checksum = index_checksum(edge_dst, edge_src);
  if (reuse) {
    // drop entries whose tensors are gone, keep at most four
    std::vector<std::shared_ptr<Topology>> alive;
    for (auto& e : cache)
      if (!e->dst_ref.expired() && !e->src_ref.expired()) alive.push_back(e);
    cache.swap(alive);
    cache.push_back(t);
    if (cache.size() > 4) cache.erase(cache.begin());
  }
  return t;
}

const Csr& by_dst(Topology& t) {
  std::lock_guard<std::mutex> lock(t.build);
  if (!t.has_dst) {
    t.by_dst = build_csr(t.dst, t.src, t.num_nodes, t.num_edges);
    t.has_dst = true;
  }
  return t.by_dst;
}

const Csr& by_src(Topology& t) {
  std::lock_guard<std::mutex> lock(t.build);
  if (!t.has_src) {
    if (t.pairing && t.has_dst && t.num_edges > 0) {
      // a paired list: row j of the by-source CSR holds the partners of the edges of row j of the dst-CSR (no sort)
      t.by_src.rowptr = t.by_dst.rowptr;
      t.by_src.other = t.by_dst.other;
      t.by_src.edge_id = at::empty_like(t.by_dst.edge_id);
      NQA_CALL(nqa_csr_from_pairs(static_cast<const int32_t*>(t.by_dst.edge_id.data_ptr()),
                                  static_cast<const int32_t*>(t.pairing->partner.data_ptr()), t.num_edges,
                                  static_cast<int32_t*>(t.by_src.edge_id.data_ptr()), stream_of(t.dst)),
               "nqa_csr_from_pairs");
    } else {
      t.by_src = build_csr(t.src, t.dst, t.num_nodes, t.num_edges);
    }
    t.has_src = true;
  }
  return t.by_src;
}

const int32_t* i32(const Tensor& t) { return static_cast<const int32_t*>(t.data_ptr()); }

// EdgeTopology.pairing (nequip_amd/nn/_topology.py): nullptr when some edge has no unique reverse partner.  One host
// synchronisation per topology entry to read the verdict.
std::shared_ptr<Pairing> pairing_of(Topology& t, const OptTensor& shift) {
  const Csr& cd = by_dst(t);  // (the reverse of (i <- j) is looked up in row j of the dst-CSR: no sort)
  std::lock_guard<std::mutex> lock(t.build);
  const void* sp = ptr(shift);
  // key as EdgeTopology._shift_key (nn/_topology.py): address, in-place version counter and size -- cell shifts rewritten in
  // place through torch (cache mode 2 keeps entries across evaluations) must not meet a stale pairing
  const uint32_t sv = sp != nullptr ? (*shift)._version() : 0u;
  const int64_t sn = sp != nullptr ? shift->numel() : -1;
  if (t.pairing_done && t.pairing_shift == sp && t.pairing_shift_version == sv && t.pairing_shift_numel == sn) return t.pairing;
  if (t.pairing_done && t.has_src) {  // the by-source CSR of a paired list was derived from the old partner map
    t.has_src = false;
    t.by_src = Csr();
  }
  t.pairing_done = true;
  t.pairing_shift = sp;
  t.pairing_shift_version = sv;
  t.pairing_shift_numel = sn;
  t.pairing_shift_hold = sp != nullptr ? *shift : Tensor();
  t.pairing.reset();
  const int64_t E = t.num_edges;
  if (E == 0 || (E & 1) != 0 || env_on("NQA_NO_PAIRED")) return nullptr;
  Tensor sh;
  int32_t sdt = NQA_F64;
  if (sp != nullptr) {
    sh = *shift;
    if (sh.scalar_type() != at::kFloat && sh.scalar_type() != at::kDouble) sh = sh.to(at::kDouble);
    sh = sh.contiguous();
    sdt = sh.scalar_type() == at::kFloat ? NQA_F32 : NQA_F64;
  }
  const auto o32 = t.dst.options().dtype(at::kInt);
  auto p = std::make_shared<Pairing>();
  p->rows = at::empty({E}, o32);
  p->partner = at::empty({E}, o32);
  p->rep = at::empty({E / 2}, t.dst.options());
  Tensor ok = at::zeros({1}, o32);
  const int64_t ws_bytes = nqa_edge_pairs_workspace_bytes(E);
  TORCH_CHECK(ws_bytes >= 0, "nequip_amd: nqa_edge_pairs_workspace_bytes failed");
  Tensor ws = at::empty({std::max<int64_t>(ws_bytes, 1)}, t.dst.options().dtype(at::kByte));
  NQA_CALL(nqa_edge_pairs(static_cast<const int64_t*>(t.dst.data_ptr()), static_cast<const int64_t*>(t.src.data_ptr()),
                          ptr(sh), sdt, i32(cd.rowptr), i32(cd.edge_id), i32(cd.other), E, t.num_nodes, ws.data_ptr(), ws_bytes,
                          static_cast<int32_t*>(p->rows.data_ptr()), static_cast<int64_t*>(p->rep.data_ptr()),
                          static_cast<int32_t*>(p->partner.data_ptr()), static_cast<int32_t*>(ok.data_ptr()), stream_of(t.dst)),
           "nqa_edge_pairs");
  if (ok.item<int32_t>() != 1) return nullptr;
  p->P = E / 2;
  t.pairing = p;
  return p;
}

const Tensor& slots_dst(Topology& t, Pairing& p) {
  const Csr& c = by_dst(t);
  std::lock_guard<std::mutex> lock(t.build);
  if (!p.slots_dst.defined()) p.slots_dst = p.rows.index_select(0, c.edge_id.slice(0, 0, t.num_edges).to(at::kLong)).contiguous();
  return p.slots_dst;
}

const Tensor& slots_src(Topology& t, Pairing& p) {
  const Csr& c = by_src(t);
  std::lock_guard<std::mutex> lock(t.build);
  if (!p.slots_src.defined()) p.slots_src = p.rows.index_select(0, c.edge_id.slice(0, 0, t.num_edges).to(at::kLong)).contiguous();
  return p.slots_src;
}

// nequip_amd/nn/_topology.py::owner_lists: the pair lists of the pair-centric backward from the pairing and the dst-CSR
// (nqa_pair_owner_lists: no sort, no synchronisation)
void owner_lists(Topology& t, Pairing& p) {
  const Csr& c = by_dst(t);
  std::lock_guard<std::mutex> lock(t.build);
  if (p.has_owner) return;
  const int64_t P = p.P, N = t.num_nodes, E = t.num_edges;
  const auto o32 = t.dst.options().dtype(at::kInt);
  p.owner[0] = at::empty({N + 1}, o32);
  p.owner[5] = at::empty({N + 1}, o32);
  for (int k : {1, 2, 3, 4, 6}) p.owner[k] = at::empty({std::max<int64_t>(P, 1)}, o32);
  const int64_t ws_bytes = nqa_pair_owner_workspace_bytes(E, N);
  TORCH_CHECK(ws_bytes >= 0, "nequip_amd: edge list exceeds the int32 index range supported by the kernels");
  Tensor ws = at::empty({std::max<int64_t>(ws_bytes, 1)}, t.dst.options().dtype(at::kByte));
  auto w = [](Tensor& v) { return static_cast<int32_t*>(v.data_ptr()); };
  NQA_CALL(nqa_pair_owner_lists(i32(p.rows), static_cast<const int64_t*>(p.rep.data_ptr()), i32(c.rowptr), i32(c.edge_id),
                                i32(c.other), E, N, ws.data_ptr(), ws_bytes, w(p.owner[0]), w(p.owner[1]), w(p.owner[2]),
                                w(p.owner[3]), w(p.owner[4]), w(p.owner[5]), w(p.owner[6]), stream_of(t.dst)),
           "nqa_pair_owner_lists");
  p.has_owner = true;
}

// ---- tp_scatter ------------------------------------------------------------------------------------------------------
void check_tp_operands(const Plan& P, const Tensor* x, const Tensor& y, const Tensor& w, int64_t N, int64_t E) {
  if (x != nullptr)
    TORCH_CHECK(x->dim() == 2 && x->size(0) == N && x->size(1) == P.dim_in1, "nequip_amd: x must be [", N, ", ", P.dim_in1, "]");
  TORCH_CHECK(y.dim() == 2 && y.size(0) == E && y.size(1) == P.dim_in2, "nequip_amd: edge_attr must be [", E, ", ", P.dim_in2, "]");
  TORCH_CHECK(w.dim() == 2 && w.size(0) == E && w.size(1) == P.weight_numel, "nequip_amd: edge_weight must be [", E, ", ",
              P.weight_numel, "]");
}

Tensor tp_scatter_fwd(const Tensor& x_, const Tensor& y_, const Tensor& w_, const Tensor& edge_dst, const Tensor& edge_src,
                      std::string plan) {
  require_gpu(x_, "tp_scatter_fwd");
  c10::DeviceGuard guard(x_.device());
  Plan& P = plan_of(plan);
  const Tensor x = x_.contiguous(), y = y_.contiguous(), w = w_.contiguous();
  const int64_t N = x.size(0), E = edge_dst.numel();
  check_tp_operands(P, &x, y, w, N, E);
  auto topo = topology_of(edge_dst, edge_src, N);
  const Csr& c = by_dst(*topo);
  Tensor out = P.out_needs_zero ? at::zeros({N, P.dim_out}, x.options()) : at::empty({N, P.dim_out}, x.options());
  Tensor image = plan_image(P, x.device());
  NQA_CALL(nqa_tp_scatter_fwd(P.handle, image.data_ptr(), nqa_dtype(x, "tp_scatter_fwd"), x.data_ptr(), y.data_ptr(),
                              w.data_ptr(), i32(c.rowptr), i32(c.edge_id), i32(c.other), out.data_ptr(), N, E,
                              stream_of(x)),
           "nqa_tp_scatter_fwd");
  return out;
}

std::tuple<Tensor, Tensor, Tensor> tp_scatter_bwd(const Tensor& g_, const Tensor& x_, const Tensor& y_, const Tensor& w_,
                                                  const Tensor& edge_dst, const Tensor& edge_src, std::string plan,
                                                  bool need_x, bool need_y, bool need_w) {
  require_gpu(x_, "tp_scatter_bwd");
  c10::DeviceGuard guard(x_.device());
  Plan& P = plan_of(plan);
  const Tensor g = g_.contiguous(), x = x_.contiguous(), y = y_.contiguous(), w = w_.contiguous();
  const int64_t N = x.size(0), E = edge_dst.numel();
  check_tp_operands(P, &x, y, w, N, E);
  TORCH_CHECK(g.dim() == 2 && g.size(0) == N && g.size(1) == P.dim_out, "nequip_amd: grad_out must be [", N, ", ", P.dim_out, "]");
  const int32_t dt = nqa_dtype(x, "tp_scatter_bwd");
  auto topo = topology_of(edge_dst, edge_src, N);
  Tensor image = plan_image(P, x.device());
  const Tensor empty = at::empty({0}, x.options());
  const auto bytes = x.options().dtype(at::kByte);
  Tensor gx = empty, gy = empty, gw = empty;
  if (need_x && need_y && need_w && P.prefer_fused_bwd && P.fused_rows_ok) {
    const int64_t ws_bytes = nqa_tp_bwd_fused_workspace_bytes(P.handle, dt, E);
    if (ws_bytes >= 0) {  // structure-specialised float32 kernels exist for this plan
      const Csr& cd = by_dst(*topo);
      const Csr& cs = by_src(*topo);
      gx = at::empty({N, P.dim_in1}, x.options());
      gw = at::empty({E, P.weight_numel}, x.options());
      gy = at::empty({E, P.dim_in2}, x.options());
      Tensor ws = at::empty({std::max<int64_t>(ws_bytes, 1)}, bytes);
      NQA_CALL(nqa_tp_scatter_bwd_fused(P.handle, image.data_ptr(), dt, x.data_ptr(), y.data_ptr(), w.data_ptr(),
                                        g.data_ptr(), i32(cd.rowptr), i32(cd.edge_id), i32(cd.other), i32(cs.rowptr),
                                        i32(cs.edge_id), gw.data_ptr(), gy.data_ptr(), gx.data_ptr(), ws.data_ptr(),
                                        ws_bytes, N, E, stream_of(x)),
               "nqa_tp_scatter_bwd_fused");
      return std::make_tuple(gx, gy, gw);
    }
  }
  if (need_x) {
    const Csr& cs = by_src(*topo);
    gx = at::empty({N, P.dim_in1}, x.options());
    NQA_CALL(nqa_tp_scatter_bwd_x(P.handle, image.data_ptr(), dt, y.data_ptr(), w.data_ptr(), g.data_ptr(), i32(cs.rowptr),
                                  i32(cs.edge_id), i32(cs.other), gx.data_ptr(), N, E, stream_of(x)),
             "nqa_tp_scatter_bwd_x");
  }
  if (need_w || need_y) {
    const Csr& cd = by_dst(*topo);
    if (need_w) gw = at::empty({E, P.weight_numel}, x.options());
    if (need_y) gy = at::empty({E, P.dim_in2}, x.options());
    int64_t ws_bytes = 0;
    Tensor ws;
    if (need_y) {
      ws_bytes = nqa_tp_bwd_edge_workspace_bytes(P.handle, dt, E);
      TORCH_CHECK(ws_bytes >= 0, "nequip_amd: nqa_tp_bwd_edge_workspace_bytes failed");
      ws = at::empty({std::max<int64_t>(ws_bytes, 1)}, bytes);
    }
    NQA_CALL(nqa_tp_scatter_bwd_edge(P.handle, image.data_ptr(), dt, x.data_ptr(), y.data_ptr(), w.data_ptr(),
                                     g.data_ptr(), i32(cd.rowptr), i32(cd.edge_id), i32(cd.other),
                                     need_w ? gw.data_ptr() : nullptr, need_y ? gy.data_ptr() : nullptr,
                                     need_y ? ws.data_ptr() : nullptr, ws_bytes, N, E, stream_of(x)),
             "nqa_tp_scatter_bwd_edge");
  }
  return std::make_tuple(gx, gy, gw);
}

// ---- edge vectors ----------------------------------------------------------------------------------------------------
Tensor as_f64(const Tensor& t) { return (t.scalar_type() == at::kDouble ? t : t.to(at::kDouble)).contiguous(); }
OptTensor as_f64(const OptTensor& t) { return (t.has_value() && t->defined()) ? OptTensor(as_f64(*t)) : OptTensor(); }
OptTensor as_i64(const OptTensor& t) {
  if (!t.has_value() || !t->defined()) return OptTensor();
  return OptTensor((t->scalar_type() == at::kLong ? *t : t->to(at::kLong)).contiguous());
}

Tensor edge_vectors(const Tensor& pos, const OptTensor& cell, const Tensor& edge_index, const OptTensor& shift,
                    const OptTensor& batch) {
  require_gpu(pos, "edge_vectors");
  c10::DeviceGuard guard(pos.device());
  TORCH_CHECK(edge_index.scalar_type() == at::kLong, "nequip_amd::edge_vectors: edge_index must be int64");
  TORCH_CHECK(edge_index.dim() == 2 && edge_index.size(0) == 2, "nequip_amd::edge_vectors: edge_index must be [2, E]");
  const Tensor pos_c = as_f64(pos);
  const int64_t E = edge_index.size(1);
  const Tensor dst = edge_index.select(0, 0).contiguous(), src = edge_index.select(0, 1).contiguous();
  const OptTensor cell_c = as_f64(cell), shift_c = as_f64(shift), batch_c = as_i64(batch);
  Tensor vec = at::empty({E, 3}, pos.options().dtype(at::kDouble));
  NQA_CALL(nqa_edge_vectors_fwd(static_cast<const double*>(pos_c.data_ptr()), static_cast<const int64_t*>(dst.data_ptr()),
                                static_cast<const int64_t*>(src.data_ptr()), static_cast<const double*>(ptr(shift_c)),
                                static_cast<const double*>(ptr(cell_c)), static_cast<const int64_t*>(ptr(batch_c)), E,
                                static_cast<double*>(vec.data_ptr()), stream_of(pos)),
           "nqa_edge_vectors_fwd");
  return vec;
}

std::tuple<Tensor, Tensor> edge_vectors_adj(const Tensor& g_vec, const Tensor& edge_index, const OptTensor& shift,
                                            const OptTensor& batch, int64_t num_nodes, int64_t num_frames,
                                            bool need_cell) {
  require_gpu(g_vec, "edge_vectors_adj");
  c10::DeviceGuard guard(g_vec.device());
  TORCH_CHECK(edge_index.scalar_type() == at::kLong && edge_index.dim() == 2 && edge_index.size(0) == 2,
              "nequip_amd::edge_vectors_adj: edge_index must be int64 [2, E]");
  const Tensor g = as_f64(g_vec);
  const OptTensor shift_c = as_f64(shift);
  auto topo = topology_of(edge_index.select(0, 0), edge_index.select(0, 1), num_nodes);
  const Csr& cd = by_dst(*topo);
  const Csr& cs = by_src(*topo);
  const auto f64 = g.options().dtype(at::kDouble);
  Tensor g_pos = at::empty({num_nodes, 3}, f64);
  Tensor part = need_cell ? at::empty({num_nodes, 9}, f64) : Tensor();
  NQA_CALL(nqa_edge_vectors_bwd(static_cast<const double*>(g.data_ptr()), static_cast<const double*>(ptr(shift_c)),
                                i32(cd.rowptr), i32(cd.edge_id), i32(cs.rowptr), i32(cs.edge_id), num_nodes, 1.0,
                                static_cast<double*>(g_pos.data_ptr()), static_cast<double*>(ptr(part)), stream_of(g)),
           "nqa_edge_vectors_bwd");
  Tensor g_cell = at::empty({0}, f64);
  if (need_cell) {
    const OptTensor batch_c = as_i64(batch);
    if (!batch_c.has_value() || num_frames == 1) {
      g_cell = part.sum(0).view({num_frames, 3, 3});
    } else if (num_frames * num_nodes > (int64_t(1) << 24)) {
      // (nqa_frame_sum scans `batch` once per frame: beyond this product the ATen scatter is the better tool -- the limit
      // of nequip_amd/nn/utils.py::_frame_kernel_ok)
      g_cell = at::zeros({num_frames, 9}, f64).index_add_(0, *batch_c, part).view({num_frames, 3, 3});
    } else {
      g_cell = at::empty({num_frames, 9}, f64);
      NQA_CALL(nqa_frame_sum(static_cast<const double*>(part.data_ptr()), static_cast<const int64_t*>(batch_c->data_ptr()),
                             num_nodes, 9, num_frames, static_cast<double*>(g_cell.data_ptr()), stream_of(g)),
               "nqa_frame_sum");
      g_cell = g_cell.view({num_frames, 3, 3});
    }
  }
  return std::make_tuple(g_pos, g_cell);
}

// ---- edge embedding --------------------------------------------------------------------------------------------------
std::tuple<Tensor, Tensor> edge_embed_fwd(const Tensor& edge_vec, const Tensor& bessel_weights, int64_t lmax, bool want_sh,
                                          bool want_emb, int64_t nb, double rmax_recip, double p, double factor, bool f32) {
  require_gpu(edge_vec, "edge_embed_fwd");
  evaluation_generation().fetch_add(1);  // top of an exported energy graph: CSRs of earlier evaluations are not reused (mode 1)
  c10::DeviceGuard guard(edge_vec.device());
  TORCH_CHECK(edge_vec.scalar_type() == at::kDouble, "nequip_amd::edge_embed_fwd: edge vectors must be float64");
  TORCH_CHECK(!want_emb || bessel_weights.scalar_type() == at::kDouble, "nequip_amd::edge_embed_fwd: bessel weights must be float64");
  const Tensor vec = edge_vec.contiguous(), bw = bessel_weights.contiguous();
  const int64_t E = vec.size(0);
  const auto opt = vec.options().dtype(f32 ? at::kFloat : at::kDouble);
  Tensor sh = want_sh ? at::empty({E, (lmax + 1) * (lmax + 1)}, opt) : at::empty({0}, opt);
  Tensor emb = want_emb ? at::empty({E, nb}, opt) : at::empty({0}, opt);
  NQA_CALL(nqa_edge_embed_fwd(f32 ? NQA_F32 : NQA_F64, (int32_t)std::max<int64_t>(lmax, 0),
                              static_cast<const double*>(vec.data_ptr()), E, rmax_recip, nullptr, (int32_t)nb,
                              static_cast<const double*>(bw.data_ptr()), p, factor, want_sh ? sh.data_ptr() : nullptr,
                              want_emb ? emb.data_ptr() : nullptr, nullptr, stream_of(vec)),
           "nqa_edge_embed_fwd");
  return std::make_tuple(sh, emb);
}

Tensor edge_embed_bwd(const Tensor& edge_vec, const Tensor& bessel_weights, const Tensor& g_sh, const Tensor& g_emb,
                      int64_t lmax, bool want_sh, bool want_emb, int64_t nb, double rmax_recip, double p, double factor,
                      bool f32) {
  require_gpu(edge_vec, "edge_embed_bwd");
  c10::DeviceGuard guard(edge_vec.device());
  TORCH_CHECK(edge_vec.scalar_type() == at::kDouble, "nequip_amd::edge_embed_bwd: edge vectors must be float64");
  const Tensor vec = edge_vec.contiguous(), bw = bessel_weights.contiguous();
  const Tensor gs = g_sh.numel() > 0 ? g_sh.contiguous() : Tensor(), ge = g_emb.numel() > 0 ? g_emb.contiguous() : Tensor();
  const auto want = f32 ? at::kFloat : at::kDouble;
  TORCH_CHECK((!gs.defined() || gs.scalar_type() == want) && (!ge.defined() || ge.scalar_type() == want),
              "nequip_amd::edge_embed_bwd: cotangent dtype does not match the model dtype");
  const int64_t E = vec.size(0);
  Tensor g_vec = at::empty({E, 3}, vec.options());
  NQA_CALL(nqa_edge_embed_bwd(f32 ? NQA_F32 : NQA_F64, (int32_t)std::max<int64_t>(lmax, 0),
                              static_cast<const double*>(vec.data_ptr()), E, rmax_recip, nullptr, (int32_t)nb,
                              static_cast<const double*>(bw.data_ptr()), p, factor, ptr(gs), ptr(ge),
                              static_cast<double*>(g_vec.data_ptr()), stream_of(vec)),
           "nqa_edge_embed_bwd");
  return g_vec;
}

// ---- radial MLP --------------------------------------------------------------------------------------------------------
int32_t mlp_mode() { return env_on("NQA_MLP_EXACT_FP32") ? NQA_MLP_FP32 : NQA_MLP_BF16X6; }

// (image, ready): the workspace of nqa_radial_mlp_fwd / _bwd for one constant `w1` -- ready once a launch has filled it
// (the prepass folds alpha1 into the image; a launch over zero rows returns before it: no entry is made for one)
std::pair<Tensor, bool> mlp_image(const Tensor& w1, double alpha1, int32_t mode, int backward, int64_t ws_bytes, int64_t rows) {
  auto fresh = [&] { return at::empty({std::max<int64_t>(ws_bytes, 1)}, w1.options().dtype(at::kByte)); };
  if (rows == 0) return {fresh(), false};
  char a[40];
  std::snprintf(a, sizeof(a), "%.17g", alpha1);
  return const_cached(w1, "mlp_image:" + std::to_string(mode) + ":" + std::to_string(backward) + ":" + a, fresh);
}

void check_mlp(const Tensor& emb, const Tensor& w0, const Tensor& w1, const char* op) {
  require_gpu(emb, op);
  TORCH_CHECK(emb.scalar_type() == at::kFloat && w0.scalar_type() == at::kFloat && w1.scalar_type() == at::kFloat,
              "nequip_amd::", op, ": float32 only");
  TORCH_CHECK(emb.dim() == 2 && w0.dim() == 2 && w1.dim() == 2 && w0.size(0) == emb.size(1) && w0.size(1) == w1.size(0),
              "nequip_amd::", op, ": shapes must be emb [E, nb], w0 [nb, H], w1 [H, W]");
  TORCH_CHECK(nqa_radial_mlp_supported(NQA_F32, (int32_t)emb.size(1), (int32_t)w1.size(0), (int32_t)w1.size(1)) == 1,
              "nequip_amd::", op, ": shape not supported by the fused kernel (nb <= 8, H in {64, 128}, W % 4 == 0)");
}

Tensor radial_mlp_fwd(const Tensor& emb_, const Tensor& w0_, const Tensor& w1_, double alpha0, double alpha1) {
  check_mlp(emb_, w0_, w1_, "radial_mlp_fwd");
  c10::DeviceGuard guard(emb_.device());
  const Tensor emb = emb_.contiguous(), w0 = w0_.contiguous(), w1 = w1_.contiguous();
  const int64_t E = emb.size(0);
  const int32_t nb = (int32_t)emb.size(1), H = (int32_t)w1.size(0), W = (int32_t)w1.size(1);
  int32_t mode = mlp_mode();
  if (mode == NQA_MLP_BF16X6 && !(std::getenv("NQA_MLP_FWD_F16") && std::getenv("NQA_MLP_FWD_F16")[0] == '0'))
    mode = NQA_MLP_F16X3;  // (nequip_amd/nn/mlp.py::forward_mode)
  Tensor out = at::empty({E, W}, emb.options());
  const int64_t ws_bytes = nqa_radial_mlp_workspace_bytes(mode, 0, H, W);
  TORCH_CHECK(ws_bytes >= 0, "nequip_amd::radial_mlp_fwd: workspace query failed");
  // the split / re-laid-out second-layer weights: built by the kernel's prepass on the first call with a constant `w1`
  const auto ws = mlp_image(w1, alpha1, mode, 0, ws_bytes, E);
  NQA_CALL(nqa_radial_mlp_fwd(NQA_F32, mode, emb.data_ptr(), w0.data_ptr(), alpha0, w1.data_ptr(), alpha1, nb, H, W, E,
                              out.data_ptr(), ws.first.data_ptr(), ws_bytes, ws.second ? 1 : 0, stream_of(emb)),
           "nqa_radial_mlp_fwd");
  return out;
}

Tensor radial_mlp_bwd(const Tensor& emb_, const Tensor& w0_, const Tensor& w1_, const Tensor& g_, double alpha0,
                      double alpha1) {
  check_mlp(emb_, w0_, w1_, "radial_mlp_bwd");
  c10::DeviceGuard guard(emb_.device());
  const Tensor emb = emb_.contiguous(), w0 = w0_.contiguous(), w1 = w1_.contiguous(), g = g_.contiguous();
  const int64_t E = emb.size(0);
  const int32_t nb = (int32_t)emb.size(1), H = (int32_t)w1.size(0), W = (int32_t)w1.size(1);
  TORCH_CHECK(g.scalar_type() == at::kFloat && g.dim() == 2 && g.size(0) == E && g.size(1) == W,
              "nequip_amd::radial_mlp_bwd: g must be float32 [E, W]");
  int32_t mode = mlp_mode();
  if (mode == NQA_MLP_BF16X6 && !(std::getenv("NQA_MLP_BWD_F16") && std::getenv("NQA_MLP_BWD_F16")[0] == '0'))
    mode = NQA_MLP_F16X3;  // (nequip_amd/nn/mlp.py::backward_mode)
  Tensor g_emb = at::empty_like(emb);
  const int64_t ws_bytes = nqa_radial_mlp_workspace_bytes(mode, 1, H, W);
  TORCH_CHECK(ws_bytes >= 0, "nequip_amd::radial_mlp_bwd: workspace query failed");
  const auto ws = mlp_image(w1, alpha1, mode, 1, ws_bytes, E);
  NQA_CALL(nqa_radial_mlp_bwd(NQA_F32, mode, emb.data_ptr(), w0.data_ptr(), alpha0, w1.data_ptr(), alpha1, g.data_ptr(), nb,
                              H, W, E, g_emb.data_ptr(), ws.first.data_ptr(), ws_bytes, ws.second ? 1 : 0, stream_of(emb)),
           "nqa_radial_mlp_bwd");
  return g_emb;
}

// gradient w.r.t. the embedding rows of the pairs from the two halves of the weight gradient (nqa_radial_mlp_bwd_paired)
Tensor radial_mlp_bwd_halves(const Tensor& emb, const Tensor& w0, const Tensor& w1, const Tensor& g1, const Tensor& g2,
                             double alpha0, double alpha1) {
  const int64_t P = emb.size(0);
  const int32_t nb = (int32_t)emb.size(1), H = (int32_t)w1.size(0), W = (int32_t)w1.size(1);
  int32_t mode = mlp_mode();
  if (mode == NQA_MLP_BF16X6 && !(std::getenv("NQA_MLP_BWD_F16") && std::getenv("NQA_MLP_BWD_F16")[0] == '0'))
    mode = NQA_MLP_F16X3;
  Tensor g_emb = at::empty_like(emb);
  const int64_t ws_bytes = nqa_radial_mlp_workspace_bytes(mode, 1, H, W);
  TORCH_CHECK(ws_bytes >= 0, "nequip_amd::radial_tp_bwd: workspace query failed");
  const auto ws = mlp_image(w1, alpha1, mode, 1, ws_bytes, P);
  NQA_CALL(nqa_radial_mlp_bwd_paired(NQA_F32, mode, emb.data_ptr(), w0.data_ptr(), alpha0, w1.data_ptr(), alpha1,
                                     g1.data_ptr(), g2.data_ptr(), nb, H, W, P, g_emb.data_ptr(), ws.first.data_ptr(),
                                     ws_bytes, ws.second ? 1 : 0, stream_of(emb)),
           "nqa_radial_mlp_bwd_paired");
  return g_emb;
}

// ---- radial_tp: radial MLP + tensor-product scatter of one convolution, pairing decided here (nn/_radial_tp_ops.py) -----
bool plan_has_spec(Plan& P) { return nqa_tp_bwd_fused_workspace_bytes(P.handle, NQA_F32, 0) >= 0; }

bool pair_backward_pays(const Tensor& g) {  // nequip_amd/nn/_paired_radial.py::_pair_backward_pays
  if (env_on("NQA_NO_PAIR_BWD")) return false;
  const char* lim = std::getenv("NQA_PAIR_BWD_MAX_MB");
  if (lim == nullptr || lim[0] == '\0') return true;
  return (double)g.numel() * (double)g.element_size() <= std::atof(lim) * (double)(1 << 20);
}

Tensor pair_rows(const Tensor& emb, Pairing& p) {
  Tensor out = at::empty({p.P, emb.size(1)}, emb.options());
  NQA_CALL(nqa_pair_gather(emb.data_ptr(), static_cast<const int64_t*>(p.rep.data_ptr()), p.P, (int32_t)emb.size(1),
                           out.data_ptr(), stream_of(emb)),
           "nqa_pair_gather");
  return out;
}

std::shared_ptr<Pairing> radial_tp_pairing(Plan& P, Topology& t, const OptTensor& shift) {
  if (!plan_has_spec(P)) return nullptr;
  return pairing_of(t, shift);
}

void check_radial_tp(Plan& P, const Tensor& emb, const Tensor& x, const Tensor& y, const Tensor& w0, const Tensor& w1,
                     int64_t E, const char* op) {
  check_mlp(emb, w0, w1, op);
  TORCH_CHECK(x.scalar_type() == at::kFloat && y.scalar_type() == at::kFloat, "nequip_amd::", op, ": float32 only");
  TORCH_CHECK(emb.size(0) == E, "nequip_amd::", op, ": one embedding row per edge");
  TORCH_CHECK(x.dim() == 2 && x.size(1) == P.dim_in1, "nequip_amd::", op, ": x must be [N, ", P.dim_in1, "]");
  TORCH_CHECK(y.dim() == 2 && y.size(0) == E && y.size(1) == P.dim_in2, "nequip_amd::", op, ": edge_attr must be [E, ", P.dim_in2, "]");
  TORCH_CHECK(w1.size(1) == P.weight_numel, "nequip_amd::", op, ": w1 must be [H, ", P.weight_numel, "]");
}

std::tuple<Tensor, Tensor> radial_tp_fwd(const Tensor& emb_, const Tensor& x_, const Tensor& y_, const Tensor& w0_,
                                         const Tensor& w1_, double alpha0, double alpha1, const Tensor& edge_dst,
                                         const Tensor& edge_src, const OptTensor& edge_shift, std::string plan) {
  require_gpu(x_, "radial_tp_fwd");
  c10::DeviceGuard guard(x_.device());
  Plan& P = plan_of(plan);
  const Tensor emb = emb_.contiguous(), x = x_.contiguous(), y = y_.contiguous(), w0 = w0_.contiguous(), w1 = w1_.contiguous();
  const int64_t N = x.size(0), E = edge_dst.numel();
  check_radial_tp(P, emb, x, y, w0, w1, E, "radial_tp_fwd");
  auto topo = topology_of(edge_dst, edge_src, N);
  auto pairing = radial_tp_pairing(P, *topo, edge_shift);
  const Csr& c = by_dst(*topo);
  Tensor out = P.out_needs_zero ? at::zeros({N, P.dim_out}, x.options()) : at::empty({N, P.dim_out}, x.options());
  Tensor image = plan_image(P, x.device());
  if (pairing) {
    const Tensor w_rows = radial_mlp_fwd(pair_rows(emb, *pairing), w0, w1, alpha0, alpha1);
    const Tensor& slots = slots_dst(*topo, *pairing);
    NQA_CALL(nqa_tp_scatter_fwd_paired(P.handle, image.data_ptr(), NQA_F32, x.data_ptr(), y.data_ptr(), w_rows.data_ptr(),
                                       i32(c.rowptr), i32(c.edge_id), i32(c.other), out.data_ptr(), N, E, i32(slots),
                                       pairing->P, stream_of(x)),
             "nqa_tp_scatter_fwd_paired");
    return std::make_tuple(out, w_rows.view({-1}));
  }
  const Tensor w = radial_mlp_fwd(emb, w0, w1, alpha0, alpha1);
  NQA_CALL(nqa_tp_scatter_fwd(P.handle, image.data_ptr(), NQA_F32, x.data_ptr(), y.data_ptr(), w.data_ptr(), i32(c.rowptr),
                              i32(c.edge_id), i32(c.other), out.data_ptr(), N, E, stream_of(x)),
           "nqa_tp_scatter_fwd");
  return std::make_tuple(out, at::empty({(E / 2) * P.weight_numel}, x.options()));
}

std::tuple<Tensor, Tensor, Tensor> radial_tp_bwd(const Tensor& g_, const Tensor& emb_, const Tensor& x_, const Tensor& y_,
                                                 const Tensor& w_rows_, const Tensor& w0_, const Tensor& w1_, double alpha0,
                                                 double alpha1, const Tensor& edge_dst, const Tensor& edge_src,
                                                 const OptTensor& edge_shift, std::string plan, bool need_emb, bool need_x,
                                                 bool need_y) {
  require_gpu(x_, "radial_tp_bwd");
  c10::DeviceGuard guard(x_.device());
  Plan& P = plan_of(plan);
  const Tensor g = g_.contiguous(), emb = emb_.contiguous(), x = x_.contiguous(), y = y_.contiguous();
  const Tensor w0 = w0_.contiguous(), w1 = w1_.contiguous();
  const int64_t N = x.size(0), E = edge_dst.numel(), W = P.weight_numel;
  check_radial_tp(P, emb, x, y, w0, w1, E, "radial_tp_bwd");
  TORCH_CHECK(g.dim() == 2 && g.size(0) == N && g.size(1) == P.dim_out && g.scalar_type() == at::kFloat,
              "nequip_amd::radial_tp_bwd: grad_out must be float32 [", N, ", ", P.dim_out, "]");
  auto topo = topology_of(edge_dst, edge_src, N);
  auto pairing = radial_tp_pairing(P, *topo, edge_shift);
  const Tensor empty = at::empty({0}, x.options());
  if (!pairing) {  // the per-edge kernels; the weight rows were not kept (w_rows is a placeholder): one more MLP forward
    const Tensor w = radial_mlp_fwd(emb, w0, w1, alpha0, alpha1);
    auto r = tp_scatter_bwd(g, x, y, w, edge_dst, edge_src, plan, need_x, need_y, need_emb);
    Tensor g_emb = need_emb ? radial_mlp_bwd(emb, w0, w1, std::get<2>(r), alpha0, alpha1) : empty;
    return std::make_tuple(g_emb, std::get<0>(r), std::get<1>(r));
  }
  Pairing& pr = *pairing;
  const int64_t Pn = pr.P;
  TORCH_CHECK(w_rows_.numel() == Pn * W, "nequip_amd::radial_tp_bwd: w_rows is not the forward's [E / 2, W] rows");
  const Tensor w_rows = w_rows_.contiguous().view({Pn, W});
  Tensor image = plan_image(P, x.device());
  const auto bytes = x.options().dtype(at::kByte);
  Tensor gx = empty, gy = empty, G;
  bool folded = false, done = false;
  // the choices of _PairedRadialTPFn.backward (nequip_amd/nn/_paired_radial.py)
  auto run_pairs = [&](bool with_gx) -> bool {
    const int64_t ws_bytes = nqa_tp_bwd_pairs_workspace_bytes(P.handle, NQA_F32, E);
    if (ws_bytes < 0) return false;
    owner_lists(*topo, pr);
    if (with_gx) gx = at::empty({N, P.dim_in1}, x.options());
    G = at::empty({Pn, W}, x.options());
    gy = at::empty({E, P.dim_in2}, x.options());
    Tensor ws = at::empty({std::max<int64_t>(ws_bytes, 1)}, bytes);
    NQA_CALL(nqa_tp_scatter_bwd_pairs(P.handle, image.data_ptr(), NQA_F32, x.data_ptr(), y.data_ptr(), w_rows.data_ptr(),
                                      g.data_ptr(), i32(pr.owner[0]), i32(pr.owner[1]), i32(pr.owner[2]), i32(pr.owner[3]),
                                      i32(pr.owner[4]), i32(pr.owner[5]), i32(pr.owner[6]), G.data_ptr(), gy.data_ptr(),
                                      with_gx ? gx.data_ptr() : nullptr, ws.data_ptr(), ws_bytes, N, E, stream_of(x)),
             "nqa_tp_scatter_bwd_pairs");
    return true;
  };
  if (need_emb && need_x && need_y && P.prefer_fused_bwd && !env_on("NQA_NO_FUSED_BWD")) {
    if (pair_backward_pays(g) && run_pairs(true)) {
      folded = done = true;
    } else if (P.fused_rows_ok) {
      const int64_t ws_bytes = nqa_tp_bwd_fused_workspace_bytes(P.handle, NQA_F32, E);
      if (ws_bytes >= 0) {
        const Csr& cd = by_dst(*topo);
        const Csr& cs = by_src(*topo);
        const Tensor& slots = slots_dst(*topo, pr);
        gx = at::empty({N, P.dim_in1}, x.options());
        G = at::empty({2 * Pn, W}, x.options());
        gy = at::empty({E, P.dim_in2}, x.options());
        Tensor ws = at::empty({std::max<int64_t>(ws_bytes, 1)}, bytes);
        NQA_CALL(nqa_tp_scatter_bwd_fused_paired(P.handle, image.data_ptr(), NQA_F32, x.data_ptr(), y.data_ptr(),
                                                 w_rows.data_ptr(), g.data_ptr(), i32(cd.rowptr), i32(cd.edge_id),
                                                 i32(cd.other), i32(cs.rowptr), i32(cs.edge_id), G.data_ptr(), gy.data_ptr(),
                                                 gx.data_ptr(), ws.data_ptr(), ws_bytes, N, E, i32(slots), Pn, stream_of(x)),
                 "nqa_tp_scatter_bwd_fused_paired");
        done = true;
      }
    }
  }
  if (!done) {
    if (need_x) {
      const Csr& cs = by_src(*topo);
      const Tensor& slots = slots_src(*topo, pr);
      gx = at::empty({N, P.dim_in1}, x.options());
      NQA_CALL(nqa_tp_scatter_bwd_x_paired(P.handle, image.data_ptr(), NQA_F32, y.data_ptr(), w_rows.data_ptr(), g.data_ptr(),
                                           i32(cs.rowptr), i32(cs.edge_id), i32(cs.other), gx.data_ptr(), N, E, i32(slots),
                                           Pn, stream_of(x)),
               "nqa_tp_scatter_bwd_x_paired");
    }
    Tensor gx_keep = gx;
    if (need_emb && need_y && pair_backward_pays(g) && run_pairs(false)) {
      folded = true;
      gx = gx_keep;
    } else if (need_emb || need_y) {
      const Csr& cd = by_dst(*topo);
      const Tensor& slots = slots_dst(*topo, pr);
      if (need_emb) G = at::empty({2 * Pn, W}, x.options());
      if (need_y) gy = at::empty({E, P.dim_in2}, x.options());
      int64_t ws_bytes = 0;
      Tensor ws;
      if (need_y) {
        ws_bytes = nqa_tp_bwd_edge_workspace_bytes(P.handle, NQA_F32, E);
        TORCH_CHECK(ws_bytes >= 0, "nequip_amd: nqa_tp_bwd_edge_workspace_bytes failed");
        ws = at::empty({std::max<int64_t>(ws_bytes, 1)}, bytes);
      }
      NQA_CALL(nqa_tp_scatter_bwd_edge_paired(P.handle, image.data_ptr(), NQA_F32, x.data_ptr(), y.data_ptr(),
                                              w_rows.data_ptr(), g.data_ptr(), i32(cd.rowptr), i32(cd.edge_id), i32(cd.other),
                                              need_emb ? G.data_ptr() : nullptr, need_y ? gy.data_ptr() : nullptr,
                                              need_y ? ws.data_ptr() : nullptr, ws_bytes, N, E, i32(slots), Pn, stream_of(x)),
               "nqa_tp_scatter_bwd_edge_paired");
    }
  }
  Tensor g_emb = empty;
  if (need_emb) {
    const Tensor emb_half = pair_rows(emb, pr);
    const Tensor g_half = folded ? radial_mlp_bwd(emb_half, w0, w1, G, alpha0, alpha1)
                                 : radial_mlp_bwd_halves(emb_half, w0, w1, G.slice(0, 0, Pn), G.slice(0, Pn, 2 * Pn), alpha0,
                                                         alpha1);
    g_emb = at::empty_like(emb);
    NQA_CALL(nqa_pair_expand(g_half.data_ptr(), i32(pr.rows), E, Pn, (int32_t)emb.size(1), g_emb.data_ptr(), stream_of(emb)),
             "nqa_pair_expand");
  }
  if (!need_x) gx = empty;
  if (!need_y) gy = empty;
  return std::make_tuple(g_emb, gx, gy);
}

// ---- force / virial tail (nequip_amd/nn/_force_ops.py) ------------------------------------------------------------------------
std::tuple<Tensor, Tensor, Tensor> force_virial(const Tensor& g_vec, const Tensor& edge_vec, const Tensor& edge_index,
                                                const OptTensor& batch, const OptTensor& cell, int64_t num_nodes,
                                                int64_t num_frames) {
  require_gpu(g_vec, "force_virial");
  c10::DeviceGuard guard(g_vec.device());
  TORCH_CHECK(edge_index.scalar_type() == at::kLong && edge_index.dim() == 2 && edge_index.size(0) == 2,
              "nequip_amd::force_virial: edge_index must be int64 [2, E]");
  const Tensor g = as_f64(g_vec), ev = as_f64(edge_vec);
  TORCH_CHECK(g.sizes() == ev.sizes() && g.dim() == 2 && g.size(1) == 3 && g.size(0) == edge_index.size(1),
              "nequip_amd::force_virial: g_vec / edge_vec must be [E, 3]");
  auto topo = topology_of(edge_index.select(0, 0), edge_index.select(0, 1), num_nodes);
  const Csr& cd = by_dst(*topo);
  const Csr& cs = by_src(*topo);
  const auto f64 = g.options().dtype(at::kDouble);
  Tensor forces = at::empty({num_nodes, 3}, f64), part = at::empty({num_nodes, 9}, f64);
  Tensor virial = at::empty({num_frames, 3, 3}, f64);
  const bool has_cell = cell.has_value() && cell->defined();
  Tensor stress = has_cell ? at::empty({num_frames, 3, 3}, f64) : at::empty({0}, f64);
  Tensor cell_c;
  if (has_cell) cell_c = as_f64(cell->reshape({-1, 3, 3}).expand({num_frames, 3, 3}));
  const OptTensor batch_c = as_i64(batch);
  // part[n] = sum_{e: centre(e) = n} edge_vec_e (x) g_e; sign -1: forces = -dE/dpos
  NQA_CALL(nqa_edge_vectors_bwd(static_cast<const double*>(g.data_ptr()), static_cast<const double*>(ev.data_ptr()),
                                i32(cd.rowptr), i32(cd.edge_id), i32(cs.rowptr), i32(cs.edge_id), num_nodes, -1.0,
                                static_cast<double*>(forces.data_ptr()), static_cast<double*>(part.data_ptr()), stream_of(g)),
           "nqa_edge_vectors_bwd");
  NQA_CALL(nqa_virial_finalize(static_cast<const double*>(part.data_ptr()), static_cast<const int64_t*>(ptr(batch_c)),
                               static_cast<const double*>(ptr(cell_c)), num_nodes, num_frames,
                               static_cast<double*>(virial.data_ptr()), has_cell ? static_cast<double*>(stress.data_ptr()) : nullptr,
                               stream_of(g)),
           "nqa_virial_finalize");
  return std::make_tuple(forces, virial, stress);
}

// ---- node_linear -------------------------------------------------------------------------------------------------------
// Tables of nqa_node_linear (nequip_amd/o3/_node_kernels.py::NodeLinearMeta): chunk = 8 int32 (out offset, d, mul_out, c0,
// instr begin, instr end, width, 0), instruction = 4 int32 (in offset, mul_in, weight offset, 0).
struct LinearMeta {
  std::vector<Ir> irreps_in, irreps_out;
  std::vector<std::pair<int, int>> instructions;
  std::vector<int32_t> w_off;
  int64_t wstride = 0, din = 0, dout = 0;
  std::vector<int32_t> chunks, instr;  // the "fwd" tables (grouped by output block)
  int32_t n_chunks = 0, n_instr = 0;
  std::map<int, Tensor> transpose_perm;  // by device: packed forward weights -> packed weights of the adjoint map
};

void build_linear_tables(LinearMeta& m) {
  const int width = 64;
  const auto in_off = irreps_offsets(m.irreps_in), out_off = irreps_offsets(m.irreps_out);
  m.din = irreps_dim(m.irreps_in);
  m.dout = irreps_dim(m.irreps_out);
  int32_t off = 0;
  for (const auto& io : m.instructions) {
    m.w_off.push_back(off);
    off += m.irreps_in[(size_t)io.first].mul * m.irreps_out[(size_t)io.second].mul;
  }
  m.wstride = off;
  for (size_t b = 0; b < m.irreps_out.size(); ++b) {
    const Ir& ir = m.irreps_out[b];
    if (ir.mul == 0) continue;
    const int32_t begin = m.n_instr;
    for (size_t k = 0; k < m.instructions.size(); ++k) {
      if ((size_t)m.instructions[k].second != b) continue;
      const size_t srcb = (size_t)m.instructions[k].first;
      const int32_t rec[4] = {in_off[srcb], m.irreps_in[srcb].mul, m.w_off[k], 0};
      m.instr.insert(m.instr.end(), rec, rec + 4);
      ++m.n_instr;
    }
    const int32_t end = m.n_instr;
    for (int32_t c0 = 0; c0 < ir.mul; c0 += width) {
      const int32_t rec[8] = {out_off[b], ir.d(), ir.mul, c0, begin, end, width, 0};
      m.chunks.insert(m.chunks.end(), rec, rec + 8);
      ++m.n_chunks;
    }
  }
}

// key = "irreps_in|irreps_out|i-o,i-o,..." (nequip_amd/o3/_node_ops.py::linear_key); transposed: the adjoint map
LinearMeta& linear_meta(const std::string& key, bool transposed) {
  static std::map<std::string, std::unique_ptr<LinearMeta>> metas;
  std::lock_guard<std::mutex> lock(registry_mutex());
  const std::string full = (transposed ? "T|" : "N|") + key;
  auto it = metas.find(full);
  if (it != metas.end()) return *it->second;
  const auto parts = split(key, '|');
  TORCH_CHECK(parts.size() == 3, "nequip_amd: malformed node_linear key");
  auto m = std::make_unique<LinearMeta>();
  auto a = parse_irreps(parts[0]), b = parse_irreps(parts[1]);
  std::vector<std::pair<int, int>> ins;
  for (const std::string& rec : split(parts[2], ',')) {
    if (rec.empty()) continue;
    const auto f = split(rec, '-');
    TORCH_CHECK(f.size() == 2, "nequip_amd: malformed node_linear instruction '", rec, "'");
    ins.emplace_back(std::stoi(f[0]), std::stoi(f[1]));
  }
  for (const auto& io : ins)
    TORCH_CHECK(io.first >= 0 && (size_t)io.first < a.size() && io.second >= 0 && (size_t)io.second < b.size(),
                "nequip_amd: node_linear instruction out of range");
  if (transposed) {
    m->irreps_in = b;
    m->irreps_out = a;
    for (const auto& io : ins) m->instructions.emplace_back(io.second, io.first);
  } else {
    m->irreps_in = a;
    m->irreps_out = b;
    m->instructions = ins;
  }
  build_linear_tables(*m);
  LinearMeta& ref = *m;
  metas.emplace(full, std::move(m));
  return ref;
}

// permutation that turns packed forward weights [T, wstride] (each instruction [mul_in, mul_out] row-major) into the
// packed weights of the adjoint map (each [mul_out, mul_in]) -- same offsets (NodeLinearMeta.transpose_weights)
Tensor transpose_perm(LinearMeta& fwd, const at::Device& device) {
  std::lock_guard<std::mutex> lock(registry_mutex());
  auto it = fwd.transpose_perm.find(device.index());
  if (it != fwd.transpose_perm.end()) return it->second;
  std::vector<int64_t> idx;
  idx.reserve((size_t)fwd.wstride);
  for (size_t k = 0; k < fwd.instructions.size(); ++k) {
    const int64_t mi = fwd.irreps_in[(size_t)fwd.instructions[k].first].mul;
    const int64_t mo = fwd.irreps_out[(size_t)fwd.instructions[k].second].mul;
    for (int64_t w = 0; w < mo; ++w)
      for (int64_t u = 0; u < mi; ++u) idx.push_back(fwd.w_off[k] + u * mo + w);
  }
  Tensor h = at::empty({(int64_t)idx.size()}, at::TensorOptions().dtype(at::kLong));
  if (!idx.empty()) std::memcpy(h.data_ptr(), idx.data(), idx.size() * sizeof(int64_t));
  Tensor d = h.to(device);
  fwd.transpose_perm.emplace(device.index(), d);
  return d;
}

// the adjoint map's packed weights from the forward map's, once per constant (nequip_amd/o3/_node_kernels.py::
// meta_transposed_weights)
Tensor transposed_weights(const Tensor& wp, const std::string& key) {
  return const_cached(wp, "node_transposed:" + key, [&] {
           return wp.index_select(1, transpose_perm(linear_meta(key, false), wp.device())).contiguous();
         }).first;
}

// fp16-split fragment image of packed weights [T, wstride] for the tables of M (nqa_node_weights_pack), once per constant
Tensor packed_weights(const Tensor& wp, LinearMeta& M, const std::string& full_key) {
  const bool f16 = !(std::getenv("NQA_NODE_F16") && std::getenv("NQA_NODE_F16")[0] == '0');
  return const_cached(wp, std::string("node_packed:") + (f16 ? "h:" : "b:") + full_key, [&] {
           const int32_t T = (int32_t)wp.size(0);
           const int64_t nbytes = nqa_node_weights_pack_bytes(M.chunks.data(), M.n_chunks, M.instr.data(), M.n_instr, T);
           TORCH_CHECK(nbytes >= 0, "nequip_amd: inconsistent node_linear tables");
           Tensor wf = at::empty({std::max<int64_t>(nbytes, 16)}, wp.options().dtype(at::kByte));
           NQA_CALL(nqa_node_weights_pack(wp.data_ptr(), M.chunks.data(), M.n_chunks, M.instr.data(), M.n_instr, T, wp.size(1),
                                          wf.data_ptr(), stream_of(wp)),
                    "nqa_node_weights_pack");
           return wf;
         }).first;
}

Tensor node_linear(const Tensor& x_, const Tensor& wp_, const OptTensor& addend_, const OptTensor& types_, std::string key,
                   double scale, bool transposed) {
  require_gpu(x_, "node_linear");
  c10::DeviceGuard guard(x_.device());
  const Tensor x = x_.contiguous();
  Tensor wp = wp_.contiguous();
  TORCH_CHECK(wp.dim() == 2 && wp.scalar_type() == x.scalar_type(), "nequip_amd::node_linear: weights must be [T, wstride] in the dtype of x");
  LinearMeta& M = linear_meta(key, transposed);
  if (transposed) wp = transposed_weights(wp, key);
  TORCH_CHECK(x.dim() == 2 && x.size(1) == M.din, "nequip_amd::node_linear: x must be [N, ", M.din, "]");
  TORCH_CHECK(wp.size(1) == M.wstride, "nequip_amd::node_linear: packed weights must be [T, ", M.wstride, "]");
  OptTensor addend, types;
  if (addend_.has_value() && addend_->defined()) {
    addend = addend_->contiguous();
    TORCH_CHECK(addend->sizes() == at::IntArrayRef({x.size(0), M.dout}) && addend->scalar_type() == x.scalar_type(),
                "nequip_amd::node_linear: addend must be [N, ", M.dout, "] in the dtype of x");
  }
  if (types_.has_value() && types_->defined()) {
    types = (types_->scalar_type() == at::kLong ? *types_ : types_->to(at::kLong)).contiguous();
    TORCH_CHECK(types->numel() == x.size(0), "nequip_amd::node_linear: one atom type per row");
  }
  const int64_t N = x.size(0), T = wp.size(0);
  TORCH_CHECK(T == 1 || types.has_value(), "nequip_amd::node_linear: per-type weights need atom types");
  Tensor out = at::empty({N, M.dout}, x.options());
  const bool packed = x.scalar_type() == at::kFloat && !env_on("NQA_NODE_EXACT_FP32") && M.n_instr > 0;
  if (packed) {
    const Tensor wf = packed_weights(wp, M, (transposed ? "T|" : "N|") + key);
    NQA_CALL(nqa_node_linear_packed(x.data_ptr(), wf.data_ptr(), ptr(addend), out.data_ptr(),
                                    static_cast<const int64_t*>(ptr(types)), M.chunks.data(), M.n_chunks, M.instr.data(),
                                    M.n_instr, (int32_t)T, (int32_t)M.din, (int32_t)M.dout, N, scale, stream_of(x)),
             "nqa_node_linear_packed");
    return out;
  }
  NQA_CALL(nqa_node_linear(nqa_dtype(x, "node_linear"), x.data_ptr(), wp.data_ptr(), ptr(addend), out.data_ptr(),
                           static_cast<const int64_t*>(ptr(types)), M.chunks.data(), M.n_chunks, M.instr.data(), M.n_instr,
                           (int32_t)T, wp.size(1), (int32_t)M.din, (int32_t)M.dout, N, scale, 64, stream_of(x)),
           "nqa_node_linear");
  return out;
}

// ---- gate ----------------------------------------------------------------------------------------------------------------
// Column tables of nqa_gate (nequip_amd/o3/_node_kernels.py::GateMeta): one 32-byte record per output (forward) / input
// (backward) column: <int32 a, b, c, e; double cst; int32 f, g>.
#pragma pack(push, 1)
struct GateRec {
  int32_t a, b, c, e;
  double cst;
  int32_t f, g;
};
#pragma pack(pop)
static_assert(sizeof(GateRec) == 32, "gate record layout");

struct GateMeta {
  int64_t ns = 0, ng = 0, din = 0, dout = 0;
  std::vector<uint8_t> fwd, bwd;
  std::map<int, std::pair<Tensor, Tensor>> dev;
};

int act_id(const std::string& name) {
  if (name == "identity") return 0;
  if (name == "silu") return 1;
  if (name == "tanh") return 2;
  TORCH_CHECK(false, "nequip_amd::gate: unknown activation '", name, "'");
}

std::vector<std::pair<int, double>> parse_acts(const std::string& text) {
  std::vector<std::pair<int, double>> out;
  for (const std::string& rec : split(text, ',')) {
    if (rec.empty()) continue;
    const size_t c = rec.find(':');
    TORCH_CHECK(c != std::string::npos, "nequip_amd::gate: malformed activation '", rec, "'");
    out.emplace_back(act_id(rec.substr(0, c)), std::stod(rec.substr(c + 1)));
  }
  return out;
}

// key = "irreps_scalars|acts|irreps_gates|acts|irreps_gated" (nequip_amd/o3/_node_ops.py::gate_key)
GateMeta& gate_meta(const std::string& key) {
  static std::map<std::string, std::unique_ptr<GateMeta>> metas;
  std::lock_guard<std::mutex> lock(registry_mutex());
  auto it = metas.find(key);
  if (it != metas.end()) return *it->second;
  const auto parts = split(key, '|');
  TORCH_CHECK(parts.size() == 5, "nequip_amd: malformed gate key");
  const auto scalars = parse_irreps(parts[0]), gates = parse_irreps(parts[2]), gated = parse_irreps(parts[4]);
  const auto act_s = parse_acts(parts[1]), act_g = parse_acts(parts[3]);
  TORCH_CHECK(act_s.size() >= scalars.size() && act_g.size() >= gates.size(), "nequip_amd::gate: one activation per irrep");
  auto m = std::make_unique<GateMeta>();
  m->ns = irreps_dim(scalars);
  m->ng = irreps_dim(gates);
  m->din = m->ns + m->ng + irreps_dim(gated);
  m->dout = m->ns + irreps_dim(gated);
  std::vector<std::pair<int, double>> col_act;
  for (size_t i = 0; i < scalars.size(); ++i)
    for (int u = 0; u < scalars[i].mul; ++u) col_act.push_back(act_s[i]);
  for (size_t i = 0; i < gates.size(); ++i)
    for (int u = 0; u < gates[i].mul; ++u) col_act.push_back(act_g[i]);
  auto rec = [](int32_t a, int32_t b, int32_t c, int32_t e, double cst, int32_t f, int32_t g) {
    GateRec r{a, b, c, e, cst, f, g};
    return r;
  };
  std::vector<GateRec> fwd((size_t)m->dout, rec(0, -1, 0, 0, 1.0, 0, 0)), bwd((size_t)m->din, rec(3, 0, 0, 0, 1.0, 0, 0));
  for (int64_t c = 0; c < m->ns; ++c) {
    fwd[(size_t)c] = rec((int32_t)c, -1, col_act[(size_t)c].first, 0, col_act[(size_t)c].second, 0, 0);
    bwd[(size_t)c] = rec(0, col_act[(size_t)c].first, (int32_t)c, 0, col_act[(size_t)c].second, 0, 0);
  }
  int64_t in_off = m->ns + m->ng, out_off = m->ns, goff = 0;
  for (const Ir& ir : gated) {
    const int d = ir.d();
    TORCH_CHECK(d <= 9, "nequip_amd::gate: gated irreps up to l = 4");
    for (int u = 0; u < ir.mul; ++u) {
      const int64_t gcol = m->ns + goff + u;
      TORCH_CHECK((size_t)gcol < col_act.size(), "nequip_amd::gate: fewer gates than gated channels");
      const auto act = col_act[(size_t)gcol];
      bwd[(size_t)gcol] = rec(1, act.first, (int32_t)(out_off + u * d), (int32_t)(in_off + u * d), act.second, d, 0);
      for (int mm = 0; mm < d; ++mm) {
        fwd[(size_t)(out_off + u * d + mm)] = rec((int32_t)(in_off + u * d + mm), (int32_t)gcol, act.first, 0, act.second, 0, 0);
        bwd[(size_t)(in_off + u * d + mm)] = rec(2, act.first, (int32_t)(out_off + u * d + mm), 0, act.second, 0, (int32_t)gcol);
      }
    }
    in_off += (int64_t)ir.mul * d;
    out_off += (int64_t)ir.mul * d;
    goff += ir.mul;
  }
  m->fwd.resize(fwd.size() * sizeof(GateRec));
  m->bwd.resize(bwd.size() * sizeof(GateRec));
  if (!fwd.empty()) std::memcpy(m->fwd.data(), fwd.data(), m->fwd.size());
  if (!bwd.empty()) std::memcpy(m->bwd.data(), bwd.data(), m->bwd.size());
  GateMeta& ref = *m;
  metas.emplace(key, std::move(m));
  return ref;
}

std::pair<Tensor, Tensor> gate_tables(GateMeta& m, const at::Device& device) {
  std::lock_guard<std::mutex> lock(registry_mutex());
  auto it = m.dev.find(device.index());
  if (it == m.dev.end())
    it = m.dev.emplace(device.index(), std::make_pair(bytes_on_device(m.fwd, device), bytes_on_device(m.bwd, device))).first;
  return it->second;
}

Tensor launch_gate(const Tensor& x, const Tensor& gout, GateMeta& m, int mode, const char* op) {
  require_gpu(x, op);
  c10::DeviceGuard guard(x.device());
  TORCH_CHECK(x.dim() == 2 && x.size(1) == m.din, "nequip_amd::", op, ": x must be [N, ", m.din, "]");
  if (gout.defined())
    TORCH_CHECK(gout.dim() == 2 && gout.size(0) == x.size(0) && gout.size(1) == m.dout && gout.scalar_type() == x.scalar_type(),
                "nequip_amd::", op, ": grad_out must be [N, ", m.dout, "] in the dtype of x");
  const auto tables = gate_tables(m, x.device());
  const int64_t N = x.size(0);
  Tensor out = at::empty({N, mode == 1 ? m.din : m.dout}, x.options());
  NQA_CALL(nqa_gate(nqa_dtype(x, op), mode, x.data_ptr(), ptr(gout), nullptr, out.data_ptr(),
                    (mode == 1 ? tables.second : tables.first).data_ptr(), (int32_t)m.din, (int32_t)m.dout, N, stream_of(x)),
           "nqa_gate");
  return out;
}

Tensor gate(const Tensor& x, std::string key) { return launch_gate(x.contiguous(), Tensor(), gate_meta(key), 0, "gate"); }

Tensor gate_bwd(const Tensor& x, const Tensor& g, std::string key) {
  return launch_gate(x.contiguous(), g.contiguous(), gate_meta(key), 1, "gate_bwd");
}

// ---- node_stage: Gate + linear_1 + typed self-connection of a layer boundary, one launch per direction (nqa_node_fused;
// nequip_amd/o3/_node_ops.py::node_stage, _node_kernels.py::_FusedNodeStageFn) ------------------------------------------------
struct GateBlocks {
  std::vector<nqa_gate_block> blocks;
  int64_t din = 0, dout = 0;
};

// GateMeta.blocks: (out_off, d, mul, val_off, gate_off, act, cst) per output block -- activated scalars, then gated irreps
GateBlocks& gate_blocks(const std::string& key) {
  static std::map<std::string, std::unique_ptr<GateBlocks>> metas;
  std::lock_guard<std::mutex> lock(registry_mutex());
  auto it = metas.find(key);
  if (it != metas.end()) return *it->second;
  const auto parts = split(key, '|');
  TORCH_CHECK(parts.size() == 5, "nequip_amd: malformed gate key");
  const auto scalars = parse_irreps(parts[0]), gates = parse_irreps(parts[2]), gated = parse_irreps(parts[4]);
  const auto act_s = parse_acts(parts[1]), act_g = parse_acts(parts[3]);
  TORCH_CHECK(act_s.size() >= scalars.size() && act_g.size() >= gated.size(), "nequip_amd::node_stage: one activation per irrep");
  auto m = std::make_unique<GateBlocks>();
  const int64_t ns = irreps_dim(scalars), ng = irreps_dim(gates);
  m->din = ns + ng + irreps_dim(gated);
  m->dout = ns + irreps_dim(gated);
  int32_t off = 0;
  for (size_t i = 0; i < scalars.size(); ++i) {
    if (scalars[i].mul > 0) m->blocks.push_back(nqa_gate_block{off, 1, scalars[i].mul, off, -1, act_s[i].first, act_s[i].second});
    off += scalars[i].mul;
  }
  int32_t b_in = (int32_t)(ns + ng), b_out = (int32_t)ns, b_gate = (int32_t)ns;
  for (size_t i = 0; i < gated.size(); ++i) {
    const Ir& ir = gated[i];
    if (ir.mul > 0) m->blocks.push_back(nqa_gate_block{b_out, ir.d(), ir.mul, b_in, b_gate, act_g[i].first, act_g[i].second});
    b_in += ir.mul * ir.d();
    b_out += ir.mul * ir.d();
    b_gate += ir.mul;
  }
  GateBlocks& ref = *m;
  metas.emplace(key, std::move(m));
  return ref;
}

// a permutation that groups the atoms by type (int32): ANY permutation gives the same results in nqa_node_fused, so the
// one-entry memo below -- keyed on the type tensor's memory -- can at worst cost speed
Tensor type_order(const Tensor& types) {
  static std::mutex mu;
  static const void* last_ptr = nullptr;
  static int64_t last_n = -1;
  static uint32_t last_version = 0;
  static Tensor last;
  std::lock_guard<std::mutex> lock(mu);
  if (last.defined() && last_ptr == types.data_ptr() && last_n == types.numel() && last_version == types._version() &&
      last.device() == types.device())
    return last;
  last = at::argsort(types.view({-1}), /*stable=*/true, 0, false).to(at::kInt).contiguous();
  last_ptr = types.data_ptr();
  last_n = types.numel();
  last_version = types._version();
  return last;
}

// (dim_out: the row length of `out` -- the map's output, or the gate's input rows when the launch ends in the gate's backward;
// 0 for an operand set that accumulates into the first one's tile)
void fill_part(nqa_node_part& p, const Tensor& x, const Tensor& packed, LinearMeta& M, int64_t n_types, void* out,
               int64_t dim_out, double scale, int32_t accumulate, GateBlocks* in_gate) {
  std::memset(&p, 0, sizeof(p));
  p.x = x.data_ptr();
  p.packed = packed.data_ptr();
  p.chunk_table = M.chunks.data();
  p.instr_table = M.instr.data();
  p.n_chunks = M.n_chunks;
  p.n_instr = M.n_instr;
  p.n_types = (int32_t)n_types;
  p.dim_in = (int32_t)x.size(1);  // (the row length of x: with an input gate, the PRE-gate rows)
  p.out = out;
  p.addend = nullptr;
  p.dim_out = (int32_t)dim_out;
  p.accumulate = accumulate;
  p.scale = scale;
  p.in_gate = in_gate != nullptr ? in_gate->blocks.data() : nullptr;
  p.n_in_gate = in_gate != nullptr ? (int32_t)in_gate->blocks.size() : 0;
}

struct StageOperands {
  Tensor h, types, wp1, wps, order;
  bool typed = false;
};

StageOperands stage_operands(const Tensor& h_, const Tensor& types_, const Tensor& wp1_, const Tensor& wps_, GateBlocks& G,
                             LinearMeta& M1, LinearMeta& MS, const char* op) {
  require_gpu(h_, op);
  StageOperands o;
  o.h = h_.contiguous();
  o.wp1 = wp1_.contiguous();
  o.wps = wps_.contiguous();
  TORCH_CHECK(o.h.scalar_type() == at::kFloat && o.wp1.scalar_type() == at::kFloat && o.wps.scalar_type() == at::kFloat,
              "nequip_amd::", op, ": float32 only");
  TORCH_CHECK(o.h.dim() == 2 && o.h.size(1) == G.din, "nequip_amd::", op, ": h must be [N, ", G.din, "] (pre-gate rows)");
  TORCH_CHECK(M1.din == G.dout && MS.din == G.dout, "nequip_amd::", op, ": linear_1 / self-connection do not take the gate's output");
  TORCH_CHECK(o.wp1.dim() == 2 && o.wp1.size(0) == 1 && o.wp1.size(1) == M1.wstride, "nequip_amd::", op, ": wp1 must be [1, ",
              M1.wstride, "]");
  TORCH_CHECK(o.wps.dim() == 2 && o.wps.size(1) == MS.wstride, "nequip_amd::", op, ": wps must be [T, ", MS.wstride, "]");
  TORCH_CHECK(!env_on("NQA_NODE_EXACT_FP32") && !(std::getenv("NQA_NODE_F16") && std::getenv("NQA_NODE_F16")[0] == '0'),
              "nequip_amd::", op, ": the fused node stage needs the default fp16-split packing");
  o.types = (types_.scalar_type() == at::kLong ? types_ : types_.to(at::kLong)).contiguous();
  TORCH_CHECK(o.types.numel() == o.h.size(0), "nequip_amd::", op, ": one atom type per row");
  o.typed = o.wps.size(0) > 1;
  const char* ord = std::getenv("NQA_NODE_TYPE_ORDER");
  if (o.typed && !(ord != nullptr && ord[0] == '0')) o.order = type_order(o.types);
  return o;
}

std::tuple<Tensor, Tensor> node_stage_fwd(const Tensor& h_, const Tensor& types_, const Tensor& wp1_, const Tensor& wps_,
                                          std::string gate_key, std::string lin_key, std::string sc_key, double scale) {
  c10::DeviceGuard guard(h_.device());
  GateBlocks& G = gate_blocks(gate_key);
  LinearMeta &M1 = linear_meta(lin_key, false), &MS = linear_meta(sc_key, false);
  StageOperands o = stage_operands(h_, types_, wp1_, wps_, G, M1, MS, "node_stage_fwd");
  const int64_t N = o.h.size(0);
  Tensor x1 = at::empty({N, M1.dout}, o.h.options()), sc = at::empty({N, MS.dout}, o.h.options());
  const Tensor p1 = packed_weights(o.wp1, M1, "N|" + lin_key), ps = packed_weights(o.wps, MS, "N|" + sc_key);
  nqa_node_part parts[2];
  fill_part(parts[0], o.h, p1, M1, 1, x1.data_ptr(), M1.dout, scale, 0, &G);
  fill_part(parts[1], o.h, ps, MS, o.wps.size(0), sc.data_ptr(), MS.dout, 1.0, 0, &G);
  NQA_CALL(nqa_node_fused(parts, 2, o.typed ? static_cast<const int64_t*>(o.types.data_ptr()) : nullptr,
                          o.order.defined() ? i32(o.order) : nullptr, N, nullptr, 0, nullptr, 0, stream_of(o.h)),
           "nqa_node_fused");
  return std::make_tuple(x1, sc);
}

Tensor node_stage_bwd(const Tensor& g_x1, const Tensor& g_sc, const Tensor& h_, const Tensor& types_, const Tensor& wp1_,
                      const Tensor& wps_, std::string gate_key, std::string lin_key, std::string sc_key, double scale) {
  c10::DeviceGuard guard(h_.device());
  GateBlocks& G = gate_blocks(gate_key);
  LinearMeta &M1 = linear_meta(lin_key, false), &MS = linear_meta(sc_key, false);
  LinearMeta &T1 = linear_meta(lin_key, true), &TS = linear_meta(sc_key, true);
  StageOperands o = stage_operands(h_, types_, wp1_, wps_, G, M1, MS, "node_stage_bwd");
  const int64_t N = o.h.size(0);
  const Tensor g1 = g_x1.contiguous(), gs = g_sc.contiguous();
  TORCH_CHECK(g1.scalar_type() == at::kFloat && g1.dim() == 2 && g1.size(0) == N && g1.size(1) == M1.dout &&
                  gs.scalar_type() == at::kFloat && gs.dim() == 2 && gs.size(0) == N && gs.size(1) == MS.dout,
              "nequip_amd::node_stage_bwd: gradients must be float32 [N, ", M1.dout, "] and [N, ", MS.dout, "]");
  // transposed weights; linear_1's scale is folded into them (both operand sets accumulate into one tile)
  Tensor w1t = transposed_weights(o.wp1, lin_key);
  if (scale != 1.0) {
    char a[40];
    std::snprintf(a, sizeof(a), "%.17g", scale);
    w1t = const_cached(w1t, std::string("node_scaled:") + a, [&] { return (w1t * scale).contiguous(); }).first;
  }
  const Tensor wst = transposed_weights(o.wps, sc_key);
  const Tensor p1 = packed_weights(w1t, T1, "T|" + lin_key), ps = packed_weights(wst, TS, "T|" + sc_key);
  Tensor gh = at::empty({N, G.din}, o.h.options());
  nqa_node_part parts[2];
  fill_part(parts[0], g1, p1, T1, 1, gh.data_ptr(), G.din, 1.0, 0, nullptr);
  fill_part(parts[1], gs, ps, TS, o.wps.size(0), nullptr, 0, 1.0, 1, nullptr);
  NQA_CALL(nqa_node_fused(parts, 2, o.typed ? static_cast<const int64_t*>(o.types.data_ptr()) : nullptr,
                          o.order.defined() ? i32(o.order) : nullptr, N, G.blocks.data(), (int32_t)G.blocks.size(),
                          o.h.data_ptr(), (int32_t)G.din, stream_of(o.h)),
           "nqa_node_fused");
  return gh;
}

// ---- energy head (nequip_amd/nn/_energy_head.py) -----------------------------------------------------------------------------
void check_head(const Tensor& h, const Tensor& w, const OptTensor& scales, const Tensor& types, const char* op) {
  require_gpu(h, op);
  TORCH_CHECK(h.scalar_type() == at::kFloat && w.scalar_type() == at::kFloat && h.dim() == 2 && w.dim() == 1 &&
                  w.size(0) == h.size(1) && h.size(1) % 4 == 0,
              "nequip_amd::", op, ": h [N, D] and w [D] float32, D a multiple of 4");
  TORCH_CHECK(types.numel() == h.size(0), "nequip_amd::", op, ": one atom type per row");
  if (scales.has_value() && scales->defined())
    TORCH_CHECK(scales->scalar_type() == at::kDouble, "nequip_amd::", op, ": scales / shifts are float64");
}

Tensor energy_head_fwd(const Tensor& h_, const Tensor& w_, const OptTensor& scales, const OptTensor& shifts,
                       const Tensor& types_, int64_t act, double cst) {
  check_head(h_, w_, scales, types_, "energy_head_fwd");
  c10::DeviceGuard guard(h_.device());
  const Tensor h = h_.contiguous(), w = w_.contiguous();
  const Tensor types = (types_.scalar_type() == at::kLong ? types_ : types_.to(at::kLong)).contiguous();
  const OptTensor sc = as_f64(scales), sh = as_f64(shifts);
  const int64_t N = h.size(0);
  Tensor e = at::empty({N, 1}, h.options().dtype(at::kDouble));
  NQA_CALL(nqa_energy_head(0, h.data_ptr(), w.data_ptr(), ptr(sc), sc.has_value() ? (int32_t)sc->numel() : 0, ptr(sh),
                           sh.has_value() ? (int32_t)sh->numel() : 0, static_cast<const int64_t*>(types.data_ptr()), nullptr,
                           e.data_ptr(), (int32_t)h.size(1), (int32_t)act, cst, N, stream_of(h)),
           "nqa_energy_head");
  return e;
}

Tensor energy_head_bwd(const Tensor& g_e, const Tensor& h_, const Tensor& w_, const OptTensor& scales, const Tensor& types_,
                       int64_t act, double cst) {
  check_head(h_, w_, scales, types_, "energy_head_bwd");
  c10::DeviceGuard guard(h_.device());
  const Tensor h = h_.contiguous(), w = w_.contiguous();
  const Tensor types = (types_.scalar_type() == at::kLong ? types_ : types_.to(at::kLong)).contiguous();
  const OptTensor sc = as_f64(scales);
  const int64_t N = h.size(0);
  const Tensor g = as_f64(g_e).view({-1});
  TORCH_CHECK(g.numel() == N, "nequip_amd::energy_head_bwd: one energy gradient per atom");
  Tensor gh = at::empty_like(h);
  NQA_CALL(nqa_energy_head(1, h.data_ptr(), w.data_ptr(), ptr(sc), sc.has_value() ? (int32_t)sc->numel() : 0, nullptr, 0,
                           static_cast<const int64_t*>(types.data_ptr()), g.data_ptr(), gh.data_ptr(), (int32_t)h.size(1),
                           (int32_t)act, cst, N, stream_of(h)),
           "nqa_energy_head");
  return gh;
}

const char* const kEdgeCfg = "int lmax, bool want_sh, bool want_emb, int nb, float rmax_recip, float p, float factor, bool f32";

bool already_registered() {
  return c10::Dispatcher::singleton().findSchema({"nequip_amd::tp_scatter_fwd", ""}).has_value();
}

}  // namespace

// Schemas: character for character those of the Python registrations.
TORCH_LIBRARY_FRAGMENT(nequip_amd, m) {
  if (already_registered()) {
    std::fprintf(stderr,
                 "[nequip_amd] torch.ops.nequip_amd.* are already defined in this process (Python registration): "
                 "libnequip_amd_torch.so leaves them alone\n");
    return;
  }
  m.def("tp_scatter_fwd(Tensor x, Tensor edge_attr, Tensor edge_weight, Tensor edge_dst, Tensor edge_src, str plan) -> Tensor");
  m.def("tp_scatter_bwd(Tensor grad_out, Tensor x, Tensor edge_attr, Tensor edge_weight, Tensor edge_dst, "
        "Tensor edge_src, str plan, bool need_x, bool need_y, bool need_w) -> (Tensor, Tensor, Tensor)");
  m.def("edge_vectors(Tensor pos, Tensor? cell, Tensor edge_index, Tensor? shift, Tensor? batch) -> Tensor");
  m.def("edge_vectors_adj(Tensor g_vec, Tensor edge_index, Tensor? shift, Tensor? batch, SymInt num_nodes, "
        "SymInt num_frames, bool need_cell) -> (Tensor, Tensor)");
  m.def((std::string("edge_embed_fwd(Tensor edge_vec, Tensor bessel_weights, ") + kEdgeCfg + ") -> (Tensor, Tensor)").c_str());
  m.def((std::string("edge_embed_bwd(Tensor edge_vec, Tensor bessel_weights, Tensor g_sh, Tensor g_emb, ") + kEdgeCfg + ") -> Tensor").c_str());
  m.def("radial_mlp_fwd(Tensor emb, Tensor w0, Tensor w1, float alpha0, float alpha1) -> Tensor");
  m.def("radial_mlp_bwd(Tensor emb, Tensor w0, Tensor w1, Tensor g, float alpha0, float alpha1) -> Tensor");
  m.def("node_linear(Tensor x, Tensor wp, Tensor? addend, Tensor? types, str key, float scale, bool transposed) -> Tensor");
  m.def("radial_tp_fwd(Tensor emb, Tensor x, Tensor edge_attr, Tensor w0, Tensor w1, float alpha0, float alpha1, "
        "Tensor edge_dst, Tensor edge_src, Tensor? edge_shift, str plan) -> (Tensor, Tensor)");
  m.def("radial_tp_bwd(Tensor grad_out, Tensor emb, Tensor x, Tensor edge_attr, Tensor w_rows, Tensor w0, Tensor w1, "
        "float alpha0, float alpha1, Tensor edge_dst, Tensor edge_src, Tensor? edge_shift, str plan, bool need_emb, "
        "bool need_x, bool need_y) -> (Tensor, Tensor, Tensor)");
  m.def("force_virial(Tensor g_vec, Tensor edge_vec, Tensor edge_index, Tensor? batch, Tensor? cell, SymInt num_nodes, "
        "SymInt num_frames) -> (Tensor, Tensor, Tensor)");
  m.def("node_stage_fwd(Tensor h, Tensor types, Tensor wp1, Tensor wps, str gate_key, str lin_key, str sc_key, "
        "float scale) -> (Tensor, Tensor)");
  m.def("node_stage_bwd(Tensor g_x1, Tensor g_sc, Tensor h, Tensor types, Tensor wp1, Tensor wps, str gate_key, "
        "str lin_key, str sc_key, float scale) -> Tensor");
  m.def("energy_head_fwd(Tensor h, Tensor w, Tensor? scales, Tensor? shifts, Tensor types, int act, float cst) "
        "-> Tensor");
  m.def("energy_head_bwd(Tensor g_e, Tensor h, Tensor w, Tensor? scales, Tensor types, int act, float cst) -> Tensor");
  m.def("gate(Tensor x, str key) -> Tensor");
  m.def("gate_bwd(Tensor x, Tensor g, str key) -> Tensor");
  m.impl("tp_scatter_fwd", c10::DispatchKey::CUDA, TORCH_FN(tp_scatter_fwd));
  m.impl("tp_scatter_bwd", c10::DispatchKey::CUDA, TORCH_FN(tp_scatter_bwd));
  m.impl("edge_vectors", c10::DispatchKey::CUDA, TORCH_FN(edge_vectors));
  m.impl("edge_vectors_adj", c10::DispatchKey::CUDA, TORCH_FN(edge_vectors_adj));
  m.impl("edge_embed_fwd", c10::DispatchKey::CUDA, TORCH_FN(edge_embed_fwd));
  m.impl("edge_embed_bwd", c10::DispatchKey::CUDA, TORCH_FN(edge_embed_bwd));
  m.impl("radial_mlp_fwd", c10::DispatchKey::CUDA, TORCH_FN(radial_mlp_fwd));
  m.impl("radial_mlp_bwd", c10::DispatchKey::CUDA, TORCH_FN(radial_mlp_bwd));
  m.impl("node_linear", c10::DispatchKey::CUDA, TORCH_FN(node_linear));
  m.impl("radial_tp_fwd", c10::DispatchKey::CUDA, TORCH_FN(radial_tp_fwd));
  m.impl("radial_tp_bwd", c10::DispatchKey::CUDA, TORCH_FN(radial_tp_bwd));
  m.impl("force_virial", c10::DispatchKey::CUDA, TORCH_FN(force_virial));
  m.impl("node_stage_fwd", c10::DispatchKey::CUDA, TORCH_FN(node_stage_fwd));
  m.impl("node_stage_bwd", c10::DispatchKey::CUDA, TORCH_FN(node_stage_bwd));
  m.impl("energy_head_fwd", c10::DispatchKey::CUDA, TORCH_FN(energy_head_fwd));
  m.impl("energy_head_bwd", c10::DispatchKey::CUDA, TORCH_FN(energy_head_bwd));
  m.impl("gate", c10::DispatchKey::CUDA, TORCH_FN(gate));
  m.impl("gate_bwd", c10::DispatchKey::CUDA, TORCH_FN(gate_bwd));
}

// ---- host-side introspection (tests: the tables built here against the Python host's, no GPU needed) -------------------
extern "C" {

int nqa_torch_topology_cache_mode(int mode) {
  if (mode < 0 || mode > 2) return cache_mode().load();
  return cache_mode().exchange(mode);
}

void nqa_torch_topology_invalidate(void) {
  std::lock_guard<std::mutex> lock(registry_mutex());
  topology_cache_entries().clear();
}

void nqa_torch_begin_evaluation(void) { evaluation_generation().fetch_add(1); }

// int32 tables of node_linear for `key`: returns the number of int32 written to each (or the required counts if too small)
int nqa_torch_linear_tables(const char* key, int transposed, int32_t* chunks, int32_t chunks_cap, int32_t* instr,
                            int32_t instr_cap, int64_t* dims /* din, dout, wstride */) {
  try {
    LinearMeta& m = linear_meta(key, transposed != 0);
    if (dims) dims[0] = m.din, dims[1] = m.dout, dims[2] = m.wstride;
    if ((int32_t)m.chunks.size() <= chunks_cap && chunks) std::memcpy(chunks, m.chunks.data(), m.chunks.size() * 4);
    if ((int32_t)m.instr.size() <= instr_cap && instr) std::memcpy(instr, m.instr.data(), m.instr.size() * 4);
    return (int)((m.chunks.size() << 16) | m.instr.size());
  } catch (const std::exception&) {
    return -1;
  }
}

int64_t nqa_torch_linear_transpose_perm(const char* key, int64_t* out, int64_t cap) {
  try {
    LinearMeta& m = linear_meta(key, false);
    Tensor p = transpose_perm(m, at::Device(at::kCPU));
    if (out && p.numel() <= cap) std::memcpy(out, p.data_ptr(), (size_t)p.numel() * 8);
    return p.numel();
  } catch (const std::exception&) {
    return -1;
  }
}

// gate column tables (bytes): which = 0 forward, 1 backward
int64_t nqa_torch_gate_table(const char* key, int which, uint8_t* out, int64_t cap, int64_t* dims /* din, dout */) {
  try {
    GateMeta& m = gate_meta(key);
    if (dims) dims[0] = m.din, dims[1] = m.dout;
    const std::vector<uint8_t>& t = which ? m.bwd : m.fwd;
    if (out && (int64_t)t.size() <= cap) std::memcpy(out, t.data(), t.size());
    return (int64_t)t.size();
  } catch (const std::exception&) {
    return -1;
  }
}

// the nqa_gate_block array of a gate key (node_stage_*: GateMeta.blocks of the Python host), as bytes
int64_t nqa_torch_gate_blocks(const char* key, uint8_t* out, int64_t cap, int64_t* dims /* din, dout */) {
  try {
    GateBlocks& g = gate_blocks(key);
    if (dims) dims[0] = g.din, dims[1] = g.dout;
    const int64_t n = (int64_t)(g.blocks.size() * sizeof(nqa_gate_block));
    if (out && n <= cap && n > 0) std::memcpy(out, g.blocks.data(), (size_t)n);
    return n;
  } catch (const std::exception&) {
    return -1;
  }
}

// (dim_in1, dim_in2, dim_out, weight_numel, out_needs_zero, prefer_fused, fused_rows_ok) of a plan text
int nqa_torch_plan_dims(const char* plan, int64_t* out7) {
  try {
    Plan& p = plan_of(plan);
    out7[0] = p.dim_in1, out7[1] = p.dim_in2, out7[2] = p.dim_out, out7[3] = p.weight_numel;
    out7[4] = p.out_needs_zero, out7[5] = p.prefer_fused_bwd, out7[6] = p.fused_rows_ok;
    return 0;
  } catch (const std::exception&) {
    return -1;
  }
}

int nqa_torch_ops_registered_here(void) { return 1; }
}
