// pcp_host_resident.hpp — ResidentGpuCStore: GpuCStore (pcp_host.hpp) whose node stays in HBM between consistency() calls.
//
// GpuCStore::consistency hands the engine host buffers (pcp_propagate): two PCIe copies of the WHOLE node per call.  A search calls
// consistency() once per node and, between two calls, changes almost nothing — Branch::commit restores a label and adds one branch
// constraint (search/branching/branch.rs:51-55), i.e. one bound of one variable.  This store keeps the node's rows (lb, ub, `active`
// words, status byte) in device memory and a host mirror of what they hold:
//   in : only the index range whose bounds differ from the mirror is copied host-to-device (8 bytes per changed variable, one range);
//   run: pcp_propagate_device from the current pair of rows into a second pair (out of place);
//   out: the status byte; the two rows only when the node did not fail, and then the output pair becomes the current one (a failed node's
//        output rows are unspecified, contract A.4: the input rows and the mirror survive it); only the variables the fixpoint narrowed
//        go through VStore::update.
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
      for (int k = 0; k < 2; ++k) {
        hip(hipMalloc((void**)&d_lb_[k], std::max<size_t>(n, 1) * sizeof(int32_t)));
        hip(hipMalloc((void**)&d_ub_[k], std::max<size_t>(n, 1) * sizeof(int32_t)));
        hip(hipMalloc((void**)&d_act_[k], words_cap_ * sizeof(uint64_t)));
      }
      hip(hipMalloc((void**)&d_status_, 8));
      m_lb_.clear();  // nothing resident yet
      m_act_.clear();
    }
    // in: the range of variables whose bounds differ from what the device's current rows (pair `cur_`) hold
    size_t lo = 0, hi = n;
    if (m_lb_.size() == n) {
      while (lo < n && vs.lbs()[lo] == m_lb_[lo] && vs.ubs()[lo] == m_ub_[lo]) ++lo;
      while (hi > lo && vs.lbs()[hi - 1] == m_lb_[hi - 1] && vs.ubs()[hi - 1] == m_ub_[hi - 1]) --hi;
    } else {
      m_lb_.assign(n, 0); m_ub_.assign(n, 0);
    }
    int32_t *lb_in = d_lb_[cur_], *ub_in = d_ub_[cur_], *lb_out = d_lb_[cur_ ^ 1], *ub_out = d_ub_[cur_ ^ 1];
    uint64_t *act_in = d_act_[cur_], *act_out = d_act_[cur_ ^ 1];
    if (hi > lo) {
      hip(hipMemcpy(lb_in + lo, vs.lbs().data() + lo, (hi - lo) * sizeof(int32_t), hipMemcpyHostToDevice));
      hip(hipMemcpy(ub_in + lo, vs.ubs().data() + lo, (hi - lo) * sizeof(int32_t), hipMemcpyHostToDevice));
      std::copy(vs.lbs().begin() + lo, vs.lbs().begin() + hi, m_lb_.begin() + lo);
      std::copy(vs.ubs().begin() + lo, vs.ubs().begin() + hi, m_ub_.begin() + lo);
      bytes_in_ += 2 * (hi - lo) * sizeof(int32_t);
    }
    if (words && act != m_act_) {  // the `active` words go up only when they differ from the device's
      hip(hipMemcpy(act_in, act.data(), words * sizeof(uint64_t), hipMemcpyHostToDevice));
      m_act_ = act;
      bytes_in_ += words * sizeof(uint64_t);
    }
    // run: OUT OF PLACE into the other pair of rows, so that a failed node (whose output rows are unspecified, contract A.4) leaves the
    // input rows — and the mirror that describes them — intact
    pcp_device_batch b{};
    b.lb_in = lb_in; b.ub_in = ub_in; b.lb_out = lb_out; b.ub_out = ub_out;
    b.active_in = words ? act_in : nullptr; b.active_out = words ? act_out : nullptr;
    b.status = d_status_;
    int32_t rc = pcp_propagate_device(ctx_, 1, &b, nullptr);
    if (rc == PCP_ERR_CONTRACT) throw Panic(pcp_last_error(ctx_));
    if (rc != PCP_OK) throw std::runtime_error(std::string("pcp_propagate_device: ") + pcp_last_error(ctx_));
    uint8_t status = 0;
    hip(hipMemcpy(&status, d_status_, 1, hipMemcpyDeviceToHost));  // (synchronises the null stream)
    bytes_out_ += 1;
    bytes_whole_ += 2 * (2 * n * sizeof(int32_t) + words * sizeof(uint64_t)) + 1;
    if (status == PCP_STATUS_HULL) throw Panic("a bound left the declared hull");
    if (status == PCP_FALSE) return status;  // nothing to fetch; the current rows still hold this node as it came in
    // out: the fixpoint; the output pair becomes the current one
    hip(hipMemcpy(m_lb_.data(), lb_out, n * sizeof(int32_t), hipMemcpyDeviceToHost));
    hip(hipMemcpy(m_ub_.data(), ub_out, n * sizeof(int32_t), hipMemcpyDeviceToHost));
    bytes_out_ += 2 * n * sizeof(int32_t);
    if (words) {
      hip(hipMemcpy(act.data(), act_out, words * sizeof(uint64_t), hipMemcpyDeviceToHost));
      m_act_ = act;
      bytes_out_ += words * sizeof(uint64_t);
    }
    cur_ ^= 1;
    for (size_t i = 0; i < n; ++i)  // MonotonicUpdate::update for the variables the fixpoint narrowed (variable/store.rs:151-166)
      if (m_lb_[i] != vs.lbs()[i] || m_ub_[i] != vs.ubs()[i]) vs.update(i, Interval(m_lb_[i], m_ub_[i]));
    return status;
  }

 private:
  void hip(hipError_t e) {
    if (e != hipSuccess) throw std::runtime_error(std::string("HIP: ") + hipGetErrorString(e));
  }
  void release() {
    for (int k = 0; k < 2; ++k) {
      if (d_lb_[k]) (void)hipFree(d_lb_[k]);
      if (d_ub_[k]) (void)hipFree(d_ub_[k]);
      if (d_act_[k]) (void)hipFree(d_act_[k]);
      d_lb_[k] = d_ub_[k] = nullptr; d_act_[k] = nullptr;
    }
    if (d_status_) (void)hipFree(d_status_);
    d_status_ = nullptr;
    n_ = (size_t)-1; words_cap_ = 0; cur_ = 0;
  }
  int32_t *d_lb_[2] = {nullptr, nullptr}, *d_ub_[2] = {nullptr, nullptr};  // two pairs of rows: [cur_] holds the node, the other takes the output
  uint64_t* d_act_[2] = {nullptr, nullptr};
  uint8_t* d_status_ = nullptr;
  int cur_ = 0;
  size_t n_ = (size_t)-1, words_cap_ = 0;
  std::vector<int32_t> m_lb_, m_ub_;  // what the current device rows hold (empty = nothing resident)
  std::vector<uint64_t> m_act_;
  uint64_t bytes_in_ = 0, bytes_out_ = 0, bytes_whole_ = 0;
};
using ResidentSpace = BasicSpace<ResidentGpuCStore>;

}  // namespace pcp_host
