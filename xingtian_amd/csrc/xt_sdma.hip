// A device -> page-locked-host copy on the SDMA ENGINE, issued through the HSA runtime the process already runs on
// (hsa_amd_memory_async_copy), synchronous for the calling thread.
//
// Why not hipMemcpyAsync: the HIP runtime PyTorch 2.10 ships (ROCm 7.0) executes every device -> host copy as a blit KERNEL
// (`__amd_rocclr_copyBuffer`; rocprofv3 --memory-copy-trace shows no MEMORY_COPY_DEVICE_TO_HOST in that process whatever the
// stream or the destination, tools/experiments/d2h_engine_probe.py), and a shader that posts megabytes of host writes holds up
// every dispatch that ENDS while they drain (tools/experiments/overlap_probe.hip: the next kernel of another stream starts
// only when the 75 us copy has finished, whatever its workgroup count, stream priority or store throttling).  The SDMA engine
// takes the same 4.2 MB in 81 us beside a kernel chain that runs 12 % slower meanwhile (same probe, /opt/rocm's runtime).
// The reference has no counterpart (its learner hands a numpy dict to a queue, xt/framework/learner.py:361-374).
#include <dlfcn.h>
#include <chrono>
#include <mutex>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include "xt_common.h"

namespace xt {
namespace {
struct HsaApi {
  decltype(&hsa_init) init = nullptr;
  decltype(&hsa_iterate_agents) iterate_agents = nullptr;
  decltype(&hsa_agent_get_info) agent_get_info = nullptr;
  decltype(&hsa_amd_pointer_info) pointer_info = nullptr;
  decltype(&hsa_amd_memory_async_copy) async_copy = nullptr;
  decltype(&hsa_signal_create) signal_create = nullptr;
  decltype(&hsa_signal_destroy) signal_destroy = nullptr;
  decltype(&hsa_signal_store_relaxed) signal_store = nullptr;
  decltype(&hsa_signal_wait_scacquire) signal_wait = nullptr;
  decltype(&hsa_signal_load_scacquire) signal_load = nullptr;
  hsa_agent_t cpu{};
  bool ok = false;
  const char* why = "not initialised";
};

hsa_status_t find_cpu(hsa_agent_t agent, void* data) {
  HsaApi* api = static_cast<HsaApi*>(data);
  hsa_device_type_t type;
  if (api->agent_get_info(agent, HSA_AGENT_INFO_DEVICE, &type) == HSA_STATUS_SUCCESS && type == HSA_DEVICE_TYPE_CPU) {
    api->cpu = agent;
    return HSA_STATUS_INFO_BREAK;
  }
  return HSA_STATUS_SUCCESS;
}

HsaApi& hsa() {
  static HsaApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    // the HSA runtime the HIP runtime of this process sits on (already loaded: same SONAME), never a second copy
    void* h = dlopen("libhsa-runtime64.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("libhsa-runtime64.so.1", RTLD_NOW);
    if (!h) { api.why = "libhsa-runtime64.so.1 is not loadable"; return; }
#define XT_HSA_SYM(field, name) api.field = reinterpret_cast<decltype(api.field)>(dlsym(h, name)); \
    if (!api.field) { api.why = "missing HSA symbol " name; return; }
    XT_HSA_SYM(init, "hsa_init") XT_HSA_SYM(iterate_agents, "hsa_iterate_agents") XT_HSA_SYM(agent_get_info, "hsa_agent_get_info")
    XT_HSA_SYM(pointer_info, "hsa_amd_pointer_info") XT_HSA_SYM(async_copy, "hsa_amd_memory_async_copy")
    XT_HSA_SYM(signal_create, "hsa_signal_create") XT_HSA_SYM(signal_destroy, "hsa_signal_destroy")
    XT_HSA_SYM(signal_store, "hsa_signal_store_relaxed") XT_HSA_SYM(signal_wait, "hsa_signal_wait_scacquire")
    XT_HSA_SYM(signal_load, "hsa_signal_load_scacquire")
#undef XT_HSA_SYM
    if (api.init() != HSA_STATUS_SUCCESS) { api.why = "hsa_init failed"; return; }      // (reference counted: HIP holds one)
    const hsa_status_t st = api.iterate_agents(find_cpu, &api);
    if (st != HSA_STATUS_INFO_BREAK) { api.why = "no CPU agent"; return; }
    api.ok = true;
  });
  return api;
}
}  // namespace

// -> 0, or an error message (static string) when this process cannot do it (the caller falls back to a stream copy)
const char* sdma_copy_d2h(void* dst_host, const void* src_dev, size_t bytes, unsigned long long* sig_handle) {
  hsa_signal_t sig_v{*sig_handle};
  hsa_signal_t* sig_io = &sig_v;
  HsaApi& api = hsa();
  if (!api.ok) return api.why;
  hsa_amd_pointer_info_t si{}, di{};
  si.size = sizeof(si); di.size = sizeof(di);
  if (api.pointer_info(const_cast<void*>(src_dev), &si, nullptr, nullptr, nullptr) != HSA_STATUS_SUCCESS ||
      si.type != HSA_EXT_POINTER_TYPE_HSA)
    return "the source is not a device allocation known to the HSA runtime";
  // the address the agents use for the page-locked range (hipHostMalloc: the host address itself; hipHostRegister: a mapping of
  // its own, which the HSA runtime may only know under THAT address -- asked from the HIP runtime then)
  char* dst = nullptr;
  if (api.pointer_info(dst_host, &di, nullptr, nullptr, nullptr) == HSA_STATUS_SUCCESS &&
      (di.type == HSA_EXT_POINTER_TYPE_LOCKED || di.type == HSA_EXT_POINTER_TYPE_HSA)) {
    dst = static_cast<char*>(di.agentBaseAddress) + (static_cast<char*>(dst_host) - static_cast<char*>(di.hostBaseAddress));
  } else {
    void* d = nullptr;
    if (hipHostGetDevicePointer(&d, dst_host, 0) != hipSuccess || !d) {
      (void)hipGetLastError();
      return "the destination is not page-locked memory (hipHostGetDevicePointer)";
    }
    dst = static_cast<char*>(d);
  }
  if (sig_io->handle == 0 && api.signal_create(1, 0, nullptr, sig_io) != HSA_STATUS_SUCCESS) return "hsa_signal_create failed";
  *sig_handle = sig_io->handle;
  api.signal_store(*sig_io, 1);
  if (api.async_copy(dst, api.cpu, src_dev, si.agentOwner, bytes, 0, nullptr, *sig_io) != HSA_STATUS_SUCCESS)
    return "hsa_amd_memory_async_copy failed";
  // (ACTIVE wait: a blocked one is woken by an interrupt ~1.5 ms after an 80 us copy -- measured; the caller is a helper thread)
  while (api.signal_wait(*sig_io, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE) >= 1) {}
  return nullptr;
}
void sdma_signal_destroy(unsigned long long* sig) {
  if (*sig && hsa().ok) hsa().signal_destroy(hsa_signal_t{*sig});
  *sig = 0;
}
}  // namespace xt

// ---- asynchronous copies with tickets (the rollout ingest's H2D: hipMemcpyAsync costs the staging thread ~20 us per message
// plus two event records; the HSA call ~5) ------------------------------------------------------------------------------------
namespace xt {
namespace {
constexpr int kDmaSlots = 64;
struct DmaSlot { hsa_signal_t sig{}; unsigned long long ticket = 0; bool busy = false; };
struct DmaPool {
  std::mutex mu;
  DmaSlot slot[kDmaSlots];
  unsigned long long next = 1;      // tickets count up from 1; ticket t lives in slot t % kDmaSlots until a later one takes it
};
DmaPool& pool() { static DmaPool p; return p; }

// the address an agent uses for page-locked host memory (see sdma_copy_d2h)
char* agent_address(HsaApi& api, void* host) {
  hsa_amd_pointer_info_t di{};
  di.size = sizeof(di);
  if (api.pointer_info(host, &di, nullptr, nullptr, nullptr) == HSA_STATUS_SUCCESS &&
      (di.type == HSA_EXT_POINTER_TYPE_LOCKED || di.type == HSA_EXT_POINTER_TYPE_HSA))
    return static_cast<char*>(di.agentBaseAddress) + (static_cast<char*>(host) - static_cast<char*>(di.hostBaseAddress));
  void* d = nullptr;
  if (hipHostGetDevicePointer(&d, host, 0) != hipSuccess || !d) { (void)hipGetLastError(); return nullptr; }
  return static_cast<char*>(d);
}
bool slot_done(HsaApi& api, DmaSlot& s) {
  if (!s.busy) return true;
  if (api.signal_load(s.sig) >= 1) return false;
  s.busy = false;
  return true;
}
}  // namespace
}  // namespace xt

int xt_dma_h2d_async(void* dst_dev, const void* src_host, int64_t bytes, uint64_t* ticket_out) {
  XT_REQUIRE(dst_dev && src_host && bytes > 0 && ticket_out, "xt_dma_h2d_async: null argument");
  xt::HsaApi& api = xt::hsa();
  XT_REQUIRE(api.ok, "xt_dma_h2d_async: %s", api.why);
  hsa_amd_pointer_info_t di{};
  di.size = sizeof(di);
  XT_REQUIRE(api.pointer_info(dst_dev, &di, nullptr, nullptr, nullptr) == HSA_STATUS_SUCCESS && di.type == HSA_EXT_POINTER_TYPE_HSA,
             "xt_dma_h2d_async: the destination is not a device allocation known to the HSA runtime");
  char* src = xt::agent_address(api, const_cast<void*>(src_host));
  XT_REQUIRE(src, "xt_dma_h2d_async: the source is not page-locked memory");
  xt::DmaPool& p = xt::pool();
  std::lock_guard<std::mutex> g(p.mu);
  const unsigned long long t = p.next;
  xt::DmaSlot& s = p.slot[t % xt::kDmaSlots];
  // (the slot's previous copy, 64 tickets ago, has to be over: waited for here, never in practice)
  while (!xt::slot_done(api, s)) __builtin_ia32_pause();
  if (s.sig.handle == 0) XT_REQUIRE(api.signal_create(1, 0, nullptr, &s.sig) == HSA_STATUS_SUCCESS, "xt_dma_h2d_async: hsa_signal_create failed");
  api.signal_store(s.sig, 1);
  XT_REQUIRE(api.async_copy(dst_dev, di.agentOwner, src, api.cpu, (size_t)bytes, 0, nullptr, s.sig) == HSA_STATUS_SUCCESS,
             "xt_dma_h2d_async: hsa_amd_memory_async_copy failed");
  s.ticket = t; s.busy = true;
  p.next = t + 1;
  *ticket_out = t;
  return 0;
}

// 0: every copy with a ticket <= `ticket` has landed; 1: not yet (timeout_ms == 0: query; < 0: no limit)
int xt_dma_wait_upto(uint64_t ticket, int32_t timeout_ms) {
  xt::HsaApi& api = xt::hsa();
  if (!api.ok || ticket == 0) return 0;
  xt::DmaPool& p = xt::pool();
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spins = 1;; ++spins) {
    bool pending = false;
    { std::lock_guard<std::mutex> g(p.mu);
      for (auto& s : p.slot)
        if (s.busy && s.ticket <= ticket && !xt::slot_done(api, s)) { pending = true; break; } }
    if (!pending) return 0;
    if (timeout_ms == 0) return 1;
    __builtin_ia32_pause();
    if ((spins & 0x3ff) == 0 && timeout_ms > 0 &&
        std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(timeout_ms)) return 1;
  }
}

// diagnostic / test entry: one synchronous SDMA copy (ABI >= 12)
int xt_sdma_copy_d2h(void* dst_host, const void* src_dev, int64_t bytes) {
  XT_REQUIRE(dst_host && src_dev && bytes > 0, "xt_sdma_copy_d2h: null argument");
  unsigned long long sig = 0;
  const char* err = xt::sdma_copy_d2h(dst_host, src_dev, (size_t)bytes, &sig);
  xt::sdma_signal_destroy(&sig);
  XT_REQUIRE(!err, "xt_sdma_copy_d2h: %s", err);
  return 0;
}
