// Native step driver for launch-bound configurations: one C call runs a whole captured trainer step.
//
// The device part of MultiAgentPPO.step (reference mappo.py:219-328: GAE, statistics, epochs x (forward, loss, backward,
// clip + optimiser)) is captured once into a hipGraph per sample signature; what was left around the replay was host
// work in Python -- one torch copy per sample leaf into the graph's static inputs, the replay call, the device->host
// copy of the loss terms and of the advantages -- several times the ~0.1 ms of kernels of the CartPole-sized
// configuration.  A plan holds the executable graph, the static input leaves and the outputs; srl_step_plan_run
// enqueues the input copies (from wherever the sample's leaves are: pinned ring, pageable numpy, device), the graph
// launch and the output copies on one stream and optionally waits, all behind a single FFI crossing.
#include <vector>

#include "srl_common.h"

namespace {
struct PlanIn {
  void* dst;
  size_t nbytes;
};
struct PlanOut {
  const void* src;
  void* host_dst;  // fixed destination, or null: given per run
  size_t nbytes;
};
struct StepPlan {
  hipGraphExec_t exec;
  std::vector<PlanIn> in;
  std::vector<PlanOut> out;
};
}  // namespace

extern "C" int srl_step_plan_create(void** plan_out, void* graph_exec) {
  SRL_CHECK_ARG(plan_out && graph_exec, "null argument");
  *plan_out = new StepPlan{(hipGraphExec_t)graph_exec, {}, {}};
  return 0;
}

extern "C" int srl_step_plan_add_input(void* plan, void* dst_device, int64_t nbytes) {
  SRL_CHECK_ARG(plan && dst_device && nbytes >= 0, "bad argument");
  static_cast<StepPlan*>(plan)->in.push_back({dst_device, (size_t)nbytes});
  return (int)static_cast<StepPlan*>(plan)->in.size() - 1;
}

extern "C" int srl_step_plan_add_output(void* plan, const void* src_device, int64_t nbytes, void* host_dst) {
  SRL_CHECK_ARG(plan && src_device && nbytes >= 0, "bad argument");
  static_cast<StepPlan*>(plan)->out.push_back({src_device, host_dst, (size_t)nbytes});
  return (int)static_cast<StepPlan*>(plan)->out.size() - 1;
}

// srcs[i]: where input i of this step lives (host or device; null = keep what the static leaf holds);
// host_dsts: per-run destinations of the outputs registered with a null one (may be null if there are none);
// sync != 0: wait for the stream (the outputs are then valid on return).
extern "C" int srl_step_plan_run(void* plan, void* stream, const void* const* srcs, int n_srcs, void* const* host_dsts,
                                 int n_dsts, int sync) {
  SRL_CHECK_ARG(plan, "null plan");
  StepPlan* p = static_cast<StepPlan*>(plan);
  SRL_CHECK_ARG(n_srcs == (int)p->in.size() && (srcs || n_srcs == 0), "input count mismatch");
  SRL_CHECK_ARG(n_dsts == (int)p->out.size() || n_dsts == 0, "output count mismatch");
  hipStream_t st = (hipStream_t)stream;
  for (size_t i = 0; i < p->in.size(); ++i)
    if (srcs[i] && p->in[i].nbytes)
      SRL_HIP_TRY(hipMemcpyAsync(p->in[i].dst, srcs[i], p->in[i].nbytes, hipMemcpyDefault, st));
  SRL_HIP_TRY(hipGraphLaunch(p->exec, st));
  for (size_t i = 0; i < p->out.size(); ++i) {
    void* dst = p->out[i].host_dst ? p->out[i].host_dst : (host_dsts && n_dsts ? host_dsts[i] : nullptr);
    if (dst && p->out[i].nbytes) SRL_HIP_TRY(hipMemcpyAsync(dst, p->out[i].src, p->out[i].nbytes, hipMemcpyDefault, st));
  }
  if (sync) SRL_HIP_TRY(hipStreamSynchronize(st));
  return 0;
}

extern "C" int srl_step_plan_destroy(void* plan) {
  delete static_cast<StepPlan*>(plan);
  return 0;
}
