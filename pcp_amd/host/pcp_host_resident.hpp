// pcp_host_resident.hpp — ResidentGpuCStore: GpuCStore (pcp_host.hpp) whose node stays in HBM between consistency() calls.
//
// GpuCStore::consistency hands the engine host buffers (pcp_propagate): two PCIe copies of the WHOLE node per call.  A search calls
// consistency() once per node and, between two calls, changes almost nothing — Branch::commit restores a label and adds one branch
// constraint (search/branching/branch.rs:51-55), i.e. one bound of one variable.  This store keeps the node's rows (lb, ub, `active`
// words, status byte) in device memory and a host mirror of what they hold:
//   in : only the index range whose bounds differ from the mirror is copied host-to-device (8 bytes per changed variable, one range);
//   run: pcp_propagate_device on the resident rows, in place;
//   out: the status byte; the two rows only when the node did not fail (a failed node's rows are unspecified, contract A.4 — the mirror
//        is dropped and the next call uploads the whole node), and only the variables the fixpoint narrowed go through VStore::update.
// Same results as GpuCStore, node for node (tests/test_gpu_parity.py::test_cpp_host_resident_store).
// The Rust form is integration/pcp-gpu-cstore/src/resident.rs (uncompiled); this file is its compiled, GPU-tested twin.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstring>

#include "pcp_host.hpp"

namespace pcp_host {

class ResidentGpuCStore : public GpuCStore {
 public:
  explicit ResidentGpuCStore(int hip_device = 0) : GpuCStore(hip_device) {}
  ~ResidentGpuCStore() override { release(); }
  // bytes moved over PCIe by consistency() so far, and what pcp_propagate would have moved for the same calls
  uint64_t bytes_in() const { return bytes_in_; }
  uint64_t bytes_out() const { return bytes_out_; }
  uint64_t bytes_whole_node() const { return bytes_whole_; }

 protected:
  uint8_t run_node(VStore& vs, std::vector<uint64_t>& act) override {
    const size_t n = vs.size(), words = act.size();
    if (n != n_ || words > words_cap_) {
      release();
      n_ = n;
      words_cap_ = std::max<size_t>(words, 16);
      hip(hipMalloc((void**)&d_lb_, std::max<size_t>(n, 1) * sizeof(int32_t)));
      hip(hipMalloc((void**)&d_ub_, std::max<size_t>(n, 1) * sizeof(int32_t)));
      hip(hipMalloc((void**)&d_act_, words_cap_ * sizeof(uint64_t)));
      hip(hipMalloc((void**)&d_status_, 8));
      m_lb_.clear();  // nothing resident yet
    }
    // in: the range of variables whose bounds differ from what the device rows hold
    size_t lo = 0, hi = n;
    if (m_lb_.size() == n) {
      while (lo < n && vs.lbs()[lo] == m_lb_[lo] && vs.ubs()[lo] == m_ub_[lo]) ++lo;
      while (hi > lo && vs.lbs()[hi - 1] == m_lb_[hi - 1] && vs.ubs()[hi - 1] == m_ub_[hi - 1]) --hi;
    }
    if (hi > lo) {
      hip(hipMemcpy(d_lb_ + lo, vs.lbs().data() + lo, (hi - lo) * sizeof(int32_t), hipMemcpyHostToDevice));
      hip(hipMemcpy(d_ub_ + lo, vs.ubs().data() + lo, (hi - lo) * sizeof(int32_t), hipMemcpyHostToDevice));
      bytes_in_ += 2 * (hi - lo) * sizeof(int32_t);
    }
    if (words) {
      hip(hipMemcpy(d_act_, act.data(), words * sizeof(uint64_t), hipMemcpyHostToDevice));
      bytes_in_ += words * sizeof(uint64_t);
    }
    pcp_device_batch b{};
    b.lb_in = d_lb_; b.ub_in = d_ub_; b.lb_out = d_lb_; b.ub_out = d_ub_;
    b.active_in = words ? d_act_ : nullptr; b.active_out = words ? d_act_ : nullptr;
    b.status = d_status_;
    int32_t rc = pcp_propagate_device(ctx_, 1, &b, nullptr);
    if (rc == PCP_ERR_CONTRACT) throw Panic(pcp_last_error(ctx_));
    if (rc != PCP_OK) throw std::runtime_error(std::string("pcp_propagate_device: ") + pcp_last_error(ctx_));
    uint8_t status = 0;
    hip(hipMemcpy(&status, d_status_, 1, hipMemcpyDeviceToHost));  // (synchronises the null stream)
    bytes_out_ += 1;
    bytes_whole_ += 2 * (2 * n * sizeof(int32_t) + words * sizeof(uint64_t)) + 1;
    if (status == PCP_STATUS_HULL) throw Panic("a bound left the declared hull");
    if (status == PCP_FALSE) {
      m_lb_.clear();  // the rows of a failed node are unspecified: nothing usable is resident
      return status;
    }
    m_lb_.resize(n); m_ub_.resize(n);
    hip(hipMemcpy(m_lb_.data(), d_lb_, n * sizeof(int32_t), hipMemcpyDeviceToHost));
    hip(hipMemcpy(m_ub_.data(), d_ub_, n * sizeof(int32_t), hipMemcpyDeviceToHost));
    bytes_out_ += 2 * n * sizeof(int32_t);
    if (words) {
      hip(hipMemcpy(act.data(), d_act_, words * sizeof(uint64_t), hipMemcpyDeviceToHost));
      bytes_out_ += words * sizeof(uint64_t);
    }
    for (size_t i = 0; i < n; ++i)  // MonotonicUpdate::update for the variables the fixpoint narrowed (variable/store.rs:151-166)
      if (m_lb_[i] != vs.lbs()[i] || m_ub_[i] != vs.ubs()[i]) vs.update(i, Interval(m_lb_[i], m_ub_[i]));
    return status;
  }

 private:
  void hip(hipError_t e) {
    if (e != hipSuccess) throw std::runtime_error(std::string("HIP: ") + hipGetErrorString(e));
  }
  void release() {
    if (d_lb_) (void)hipFree(d_lb_);
    if (d_ub_) (void)hipFree(d_ub_);
    if (d_act_) (void)hipFree(d_act_);
    if (d_status_) (void)hipFree(d_status_);
    d_lb_ = d_ub_ = nullptr; d_act_ = nullptr; d_status_ = nullptr;
    n_ = (size_t)-1; words_cap_ = 0;
  }
  int32_t *d_lb_ = nullptr, *d_ub_ = nullptr;
  uint64_t* d_act_ = nullptr;
  uint8_t* d_status_ = nullptr;
  size_t n_ = (size_t)-1, words_cap_ = 0;
  std::vector<int32_t> m_lb_, m_ub_;  // what the device rows hold (empty = nothing usable)
  uint64_t bytes_in_ = 0, bytes_out_ = 0, bytes_whole_ = 0;
};
using ResidentSpace = BasicSpace<ResidentGpuCStore>;

}  // namespace pcp_host
