// aqlq.cpp - libafldm_aql.so: AQL-level view of the denoise step (round 5).
//
// The step is ~220 dependent kernel launches replayed from a HIP graph; the runtime emits every one of them as an AQL
// kernel-dispatch packet with the BARRIER bit set and agent-scope acquire / release fences, whatever the data flow
// between the two kernels is.  HIP has no interface for the packet header (`hipExtAnyOrderLaunch` is not honoured on
// gfx9 and is lost in graph capture), so this library sits on the ROCr tools interface (HSA_TOOLS_LIB): every queue
// the process creates becomes an intercept queue, and the dispatch packets that pass through it can be
//   * counted and recorded (kernel object, grid, LDS bytes, kernarg address, header) - tools/aql_probe.py;
//   * re-headed according to a per-launch policy the host arms for the next N dispatches (bit 0: no barrier bit -
//     the packet may start while the packets in front of it still run; bit 1: no acquire fence; bit 2: no release fence).
// Nothing here touches a kernel's arguments or its code.  Host side: afldm_amd/aql.py.
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <hsa/hsa_api_trace.h>

#include <atomic>
#include <cstdint>
#include <cstring>
#include <mutex>

namespace {

decltype(hsa_queue_create)* g_real_create = nullptr;
decltype(hsa_amd_queue_intercept_create)* g_icreate = nullptr;
decltype(hsa_amd_queue_intercept_register)* g_iregister = nullptr;

struct Rec {                     // one dispatch packet as it arrived (56 bytes; afldm_amd/aql.py mirrors this)
  uint16_t header, setup;
  uint16_t wg[3];
  uint16_t pad;
  uint32_t grid[3];
  uint32_t priv_bytes, group_bytes;
  uint64_t kernel_object, kernarg, completion;
};
static_assert(sizeof(Rec) == 56, "record layout");

constexpr int MAXREC = 1 << 15, MAXPOL = 1 << 13;
std::mutex g_mu;
Rec g_rec[MAXREC];
int g_nrec = 0, g_recording = 0;
uint64_t g_ndispatch = 0, g_nother = 0, g_nqueues = 0, g_nrewritten = 0;
uint8_t g_pol[MAXPOL];
int g_pn = 0;
long long g_pleft = 0, g_ppos = 0;
int g_loaded = 0;

constexpr uint16_t BARRIER_BIT = 1u << HSA_PACKET_HEADER_BARRIER;
constexpr uint16_t ACQ_MASK = 3u << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE;
constexpr uint16_t REL_MASK = 3u << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE;

void handler(const void* pkts, uint64_t count, uint64_t, void*, hsa_amd_queue_intercept_packet_writer writer) {
  const hsa_kernel_dispatch_packet_t* in = static_cast<const hsa_kernel_dispatch_packet_t*>(pkts);
  hsa_kernel_dispatch_packet_t buf[32];
  std::lock_guard<std::mutex> lk(g_mu);
  uint64_t done = 0;
  while (done < count) {
    const uint64_t n = count - done < 32 ? count - done : 32;
    std::memcpy(buf, in + done, n * sizeof(buf[0]));
    for (uint64_t i = 0; i < n; ++i) {
      const unsigned type = (buf[i].header >> HSA_PACKET_HEADER_TYPE) & 0xff;
      if (type != HSA_PACKET_TYPE_KERNEL_DISPATCH) {
        ++g_nother;
        continue;
      }
      ++g_ndispatch;
      if (g_recording && g_nrec < MAXREC) {
        Rec& r = g_rec[g_nrec++];
        r.header = buf[i].header;
        r.setup = buf[i].setup;
        r.wg[0] = buf[i].workgroup_size_x; r.wg[1] = buf[i].workgroup_size_y; r.wg[2] = buf[i].workgroup_size_z;
        r.pad = 0;
        r.grid[0] = buf[i].grid_size_x; r.grid[1] = buf[i].grid_size_y; r.grid[2] = buf[i].grid_size_z;
        r.priv_bytes = buf[i].private_segment_size;
        r.group_bytes = buf[i].group_segment_size;
        r.kernel_object = buf[i].kernel_object;
        r.kernarg = reinterpret_cast<uint64_t>(buf[i].kernarg_address);
        r.completion = buf[i].completion_signal.handle;
      }
      if (g_pleft > 0 && g_pn > 0) {
        const uint8_t pol = g_pol[g_ppos % g_pn];
        ++g_ppos;
        --g_pleft;
        uint16_t h = buf[i].header;
        if (pol & 1) h &= (uint16_t)~BARRIER_BIT;
        if (pol & 2) h &= (uint16_t)~ACQ_MASK;
        if (pol & 4) h &= (uint16_t)~REL_MASK;
        if (h != buf[i].header) {
          buf[i].header = h;
          ++g_nrewritten;
        }
      }
    }
    writer(buf, n);
    done += n;
  }
}

hsa_status_t queue_create(hsa_agent_t agent, uint32_t size, hsa_queue_type32_t type,
                          void (*callback)(hsa_status_t, hsa_queue_t*, void*), void* data, uint32_t private_segment_size,
                          uint32_t group_segment_size, hsa_queue_t** queue) {
  hsa_device_type_t dt = HSA_DEVICE_TYPE_CPU;
  hsa_agent_get_info(agent, HSA_AGENT_INFO_DEVICE, &dt);
  if (dt != HSA_DEVICE_TYPE_GPU) return g_real_create(agent, size, type, callback, data, private_segment_size, group_segment_size, queue);
  hsa_status_t st = g_icreate(agent, size, type, callback, data, private_segment_size, group_segment_size, queue);
  if (st != HSA_STATUS_SUCCESS) return st;
  st = g_iregister(*queue, handler, nullptr);
  if (st == HSA_STATUS_SUCCESS) {
    std::lock_guard<std::mutex> lk(g_mu);
    ++g_nqueues;
  }
  return st;
}

}  // namespace

extern "C" {

// ROCr tools entry points
__attribute__((visibility("default"))) bool OnLoad(HsaApiTable* table, uint64_t, uint64_t, const char* const*) {
  g_real_create = table->core_->hsa_queue_create_fn;
  g_icreate = table->amd_ext_->hsa_amd_queue_intercept_create_fn;
  g_iregister = table->amd_ext_->hsa_amd_queue_intercept_register_fn;
  if (!g_real_create || !g_icreate || !g_iregister) return false;
  table->core_->hsa_queue_create_fn = queue_create;
  g_loaded = 1;
  return true;
}
__attribute__((visibility("default"))) void OnUnload() {}

// host interface (ctypes; afldm_amd/aql.py)
__attribute__((visibility("default"))) int afldm_aql_loaded() { return g_loaded; }
__attribute__((visibility("default"))) void afldm_aql_counts(uint64_t* out4) {
  std::lock_guard<std::mutex> lk(g_mu);
  out4[0] = g_ndispatch; out4[1] = g_nother; out4[2] = g_nqueues; out4[3] = g_nrewritten;
}
__attribute__((visibility("default"))) void afldm_aql_record(int on) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_recording = on;
  if (on) g_nrec = 0;
}
__attribute__((visibility("default"))) int afldm_aql_records(void* out, int max) {
  std::lock_guard<std::mutex> lk(g_mu);
  const int n = g_nrec < max ? g_nrec : max;
  std::memcpy(out, g_rec, (size_t)n * sizeof(Rec));
  return n;
}
// the next `total` dispatch packets get policy[i % n] (i counted from this call); n = 0 or total = 0 disarms
__attribute__((visibility("default"))) int afldm_aql_arm(const uint8_t* policy, int n, long long total) {
  if (n < 0 || n > MAXPOL) return -1;
  std::lock_guard<std::mutex> lk(g_mu);
  if (n > 0) std::memcpy(g_pol, policy, (size_t)n);
  g_pn = n;
  g_pleft = n > 0 ? total : 0;
  g_ppos = 0;
  return 0;
}
__attribute__((visibility("default"))) long long afldm_aql_armed_left() {
  std::lock_guard<std::mutex> lk(g_mu);
  return g_pleft;
}

}  // extern "C"
