// Stand-alone reproduction attempt of the HIP-runtime crash met under hipGraph capture / destroy churn (EXPERIMENTS.md R4-4d):
// no torch, no library of this repo -- a two-stream step graph of ~50 small kernel nodes (the shape of one sampler step: a main
// chain with two forks and two joins per "layer"), captured, instantiated, replayed in pieces with an event-synchronised drain
// to pinned memory between the pieces, and destroyed in one of three orders:
//   mode 0  "older first"  : a random OLDER executable graph is destroyed while newer ones stay alive, then the next is captured
//   mode 1  "one by one"   : every finished graph is destroyed at once (others -- the cached ones -- stay alive)
//   mode 2  "parked"       : finished graphs are parked and destroyed all together, newest first, once 12 have piled up
//                            (what decompdiff_amd/model.py does since round 4)
//   mode 3  "one by one + event churn": mode 1, and while a graph is being captured unrelated events are created, queried and destroyed (what a host framework's allocator and garbage collector do
//                            at arbitrary points: torch's pinned-memory allocator queries its events on every allocation)
// Every mode runs in a forked child (a crash of the runtime is a signal in the child, not the end of the experiment).
//   hipcc --offload-arch=gfx950 -O2 tools/graph_churn_repro.hip -o tools/_build/graph_churn_repro && tools/_build/graph_churn_repro [iterations] [repeats]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <sys/wait.h>
#include <unistd.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %d (%s) at line %d\n", (int)e_, hipGetErrorString(e_), __LINE__); _exit(3); } } while (0)

__global__ void k_work(float* p, int n, float a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = p[i] * 0.999f + a;
}

struct G { hipGraph_t g = nullptr; hipGraphExec_t e = nullptr; int id = 0; };

static hipStream_t g_churn_stream = nullptr;            // mode 3: event traffic on another stream in the middle of a capture
static void event_churn() {
  if (!g_churn_stream) return;
  hipEvent_t e[4];
  // (created, queried and destroyed only: RECORDING on another stream from the capturing thread is an "unsafe call" that
  //  invalidates a thread-local capture with error 901 -- the first version of this mode found that, not a crash)
  for (auto& x : e) { CK(hipEventCreateWithFlags(&x, hipEventDisableTiming)); }
  for (auto& x : e) { (void)hipEventQuery(x); }
  for (auto& x : e) CK(hipEventDestroy(x));
}

static G capture(hipStream_t s0, hipStream_t s1, hipEvent_t* ev, float* buf, int n, int id) {
  G out; out.id = id;
  const int nb = (n + 255) / 256;
  CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
  for (int l = 0; l < 6; ++l) {
    hipLaunchKernelGGL(k_work, dim3(nb), dim3(256), 0, s0, buf, n, 0.001f * (l + 1));                 // "node attention"
    CK(hipEventRecord(ev[4 * l], s0)); CK(hipStreamWaitEvent(s1, ev[4 * l], 0));                      // fork 1
    hipLaunchKernelGGL(k_work, dim3(nb), dim3(256), 0, s1, buf + n, n, 0.002f);
    hipLaunchKernelGGL(k_work, dim3(nb), dim3(256), 0, s0, buf + 2 * n, n, 0.003f);                   // "lin_node"
    CK(hipEventRecord(ev[4 * l + 1], s0)); CK(hipStreamWaitEvent(s1, ev[4 * l + 1], 0));              // fork 2
    hipLaunchKernelGGL(k_work, dim3(nb), dim3(256), 0, s1, buf + 3 * n, n, 0.004f);
    CK(hipEventRecord(ev[4 * l + 2], s1));
    hipLaunchKernelGGL(k_work, dim3(nb), dim3(256), 0, s1, buf + 4 * n, n, 0.005f);
    CK(hipEventRecord(ev[4 * l + 3], s1));
    event_churn();
    hipLaunchKernelGGL(k_work, dim3(nb), dim3(256), 0, s0, buf + 2 * n, n, 0.006f);                   // projections
    hipLaunchKernelGGL(k_work, dim3(nb / 2 + 1), dim3(256), 0, s0, buf + 5 * n, n / 2, 0.007f);       // "coordinate attention"
    CK(hipStreamWaitEvent(s0, ev[4 * l + 2], 0));                                                     // join 1
    hipLaunchKernelGGL(k_work, dim3(nb), dim3(256), 0, s0, buf + 3 * n, n, 0.008f);                   // "assemble"
    CK(hipStreamWaitEvent(s0, ev[4 * l + 3], 0));                                                     // join 2
  }
  hipLaunchKernelGGL(k_work, dim3(nb), dim3(256), 0, s0, buf, n, 0.009f);                             // "step"
  CK(hipStreamEndCapture(s0, &out.g));
  CK(hipGraphInstantiate(&out.e, out.g, nullptr, nullptr, 0));
  return out;
}
static void destroy(G& g) { if (g.e) CK(hipGraphExecDestroy(g.e)); if (g.g) CK(hipGraphDestroy(g.g)); g.e = nullptr; g.g = nullptr; }

static int run_mode(int mode, int iters, unsigned seed) {
  hipStream_t s0, s1, sc;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
  hipEvent_t ev[24];
  for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  const int NMAX = 1 << 16;
  float* buf; CK(hipMalloc(&buf, sizeof(float) * 6 * NMAX)); CK(hipMemset(buf, 0, sizeof(float) * 6 * NMAX));
  float* host; CK(hipHostMalloc(&host, sizeof(float) * NMAX));
  std::vector<G> cached, parked;
  if (mode == 3) { g_churn_stream = sc; mode = 1; }
  srand(seed);
  for (int it = 0; it < iters; ++it) {
    const int n = 4096 + 512 * (rand() % 64);                       // another "shape" every time
    G g = capture(s0, s1, ev, buf, n, it);
    for (int piece = 0; piece < 3; ++piece) {                       // replay in pieces, drain each behind an event
      for (int r = 0; r < 8; ++r) CK(hipGraphLaunch(g.e, s0));
      hipEvent_t done, copied;
      CK(hipEventCreateWithFlags(&done, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&copied, hipEventDisableTiming));
      CK(hipEventRecord(done, s0)); CK(hipStreamWaitEvent(sc, done, 0));
      CK(hipMemcpyAsync(host, buf, sizeof(float) * n, hipMemcpyDeviceToHost, sc));
      CK(hipEventRecord(copied, sc)); CK(hipEventSynchronize(copied));
      CK(hipEventDestroy(done)); CK(hipEventDestroy(copied));
    }
    CK(hipStreamSynchronize(s0)); CK(hipStreamSynchronize(sc));
    const bool keep = (it % 3 == 0) && cached.size() < 4;            // some chains are "cached" and stay alive
    if (keep) { cached.push_back(g); continue; }
    if (mode == 0) {                                                // an OLDER graph goes, this one stays
      cached.push_back(g);
      const size_t victim = rand() % (cached.size() - 1 ? cached.size() - 1 : 1);
      destroy(cached[victim]); cached.erase(cached.begin() + victim);
    } else if (mode == 1) {
      destroy(g);
    } else {
      parked.push_back(g);
      if (parked.size() >= 12) {                                    // all live graphs, newest first
        std::vector<G> all = cached; all.insert(all.end(), parked.begin(), parked.end());
        for (size_t i = 0; i < all.size(); ++i) for (size_t j = i + 1; j < all.size(); ++j) if (all[j].id > all[i].id) std::swap(all[i], all[j]);
        for (auto& x : all) destroy(x);
        cached.clear(); parked.clear();
      }
    }
  }
  CK(hipDeviceSynchronize());
  return 0;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 200, reps = argc > 2 ? atoi(argv[2]) : 8;
  const char* names[4] = {"older first", "one by one", "parked, all together newest first", "one by one + event churn during capture"};
  // (the parent never touches the HIP runtime: a fork after its initialisation leaves the child with a dead KFD handle)
  printf("%d repeats of %d capture/replay/destroy iterations per mode\n", reps, iters);
  for (int mode = 0; mode < 4; ++mode) {
    int crashed = 0, failed = 0;
    for (int r = 0; r < reps; ++r) {
      fflush(stdout);
      const pid_t pid = fork();
      if (pid == 0) {
        if (mode == 0 && r == 0) { int rt = 0; (void)hipRuntimeGetVersion(&rt); printf("HIP runtime version %d\n", rt); fflush(stdout); }
        _exit(run_mode(mode, iters, 1234u + 77u * r));
      }
      int st = 0; waitpid(pid, &st, 0);
      if (WIFSIGNALED(st)) { ++crashed; printf("  mode %d repeat %d: killed by signal %d\n", mode, r, WTERMSIG(st)); }
      else if (WEXITSTATUS(st) != 0) { ++failed; printf("  mode %d repeat %d: exit status %d\n", mode, r, WEXITSTATUS(st)); }
    }
    printf("mode %d (%s): %d crashed, %d failed, %d clean of %d\n", mode, names[mode], crashed, failed, reps - crashed - failed, reps);
  }
  return 0;
}
