// tde_torch_ext.cpp — the PyTorch-ROCm C++ extension of the env step path (BASELINE.json north_star: "hand-written CDNA4
// HIP kernels behind a PyTorch-ROCm C++ extension").  It is a thin binding of the SAME C-ABI entry points that
// include/tde_hip.h declares (libtde_hip.so): torch tensors in, TORCH_CHECK on dtype / shape / device / contiguity, the
// launch stream taken from torch's current HIP stream in C++, HIP errors raised as RuntimeError.  No kernel lives here.
// What it removes is the per-call ctypes marshalling of the Python binding (struct copies, byref, argument conversion:
// about 7 us per call on the host), which is what bounds closed-loop stepping (one tde_env_step per policy action,
// reference loop: gym_env.py:453-461).  The ctypes binding stays as the reference-side stub.
#include <torch/extension.h>

// PyTorch-ROCm keeps the device type "cuda" for HIP devices: the guard / stream accessors are the "masquerading" ones
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>

#include <cstring>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "tde_hip.h"

namespace {

void check_rc(int rc, const char *what)
{
    if (rc != 0) throw std::runtime_error(std::string(what) + ": " + tde_last_error());
}

const void *dev_ptr(const at::Tensor &t, at::ScalarType dt, int64_t numel, const char *name, const at::Device &dev)
{
    TORCH_CHECK(t.is_cuda(), name, " must live on a HIP device (there is no CPU path)");
    TORCH_CHECK(t.device() == dev, name, " is on ", t.device(), ", expected ", dev);
    TORCH_CHECK(t.scalar_type() == dt, name, " must be ", dt, ", got ", t.scalar_type());
    TORCH_CHECK(t.is_contiguous(), name, " must be contiguous");
    TORCH_CHECK(numel < 0 || t.numel() == numel, name, " must have ", numel, " elements, got ", t.numel());
    return t.data_ptr();
}

// ---- typed carriers of the C-ABI's argument structs -----------------------------------------------------------------
// Round 4: the extension no longer takes raw addresses.  A Config is built from the BYTES of a tde_config (length checked),
// a World and an EnvHandle from NAMED device tensors (dtype / device / contiguity / size checked against the struct's
// field), and both keep those tensors alive for as long as they exist: nothing here dereferences an integer that Python
// passed, and a state buffer that Python re-allocates cannot leave the handle pointing at freed memory (the handle
// still holds the old tensor; make a new handle for the new buffers).
class Config {
  public:
    explicit Config(const py::bytes &raw)
    {
        const std::string b = raw;
        TORCH_CHECK(b.size() == sizeof(tde_config), "tde_config is ", sizeof(tde_config), " bytes, got ", b.size());
        std::memcpy(&c, b.data(), sizeof(c));
    }
    tde_config c;
};

struct Field {
    const char *name;
    size_t off;                 // offset of the pointer member in the struct
    at::ScalarType dt;
    int kind;                   // how many elements: see count_of
    bool required;
};
enum { K_ANY = 0, K_AGENT, K_ENV, K_ENV2, K_ENV4, K_ENV8, K_SLOT8, K_ACT };

#define WF(n, dt) {#n, offsetof(tde_world, n), dt, K_ANY, true}
const Field kWorldFields[] = {WF(maps, at::kByte), WF(tri, at::kFloat), WF(cell_word, at::kUInt32), WF(cell_tri, at::kFloat),
                              WF(cell_cls2, at::kUInt32), WF(cell_sub, at::kUInt32), WF(cell_coarse, at::kByte), WF(tile_near, at::kUInt32), WF(scn, at::kByte),
                              WF(wp_xy, at::kDouble), WF(spawn, at::kByte), WF(route_xy, at::kFloat), WF(replay_states, at::kFloat),
                              WF(stoplines, at::kByte), WF(phases, at::kByte), WF(start_psi, at::kFloat), WF(first_gap, at::kUInt32)};
#undef WF
#define SF(n, dt, k, req) {#n, offsetof(tde_state, n), dt, k, req}
const Field kStateFields[] = {
    SF(x, at::kFloat, K_AGENT, true), SF(y, at::kFloat, K_AGENT, true), SF(psi, at::kFloat, K_AGENT, true), SF(v, at::kFloat, K_AGENT, true),
    SF(len, at::kFloat, K_AGENT, true), SF(wid, at::kFloat, K_AGENT, true), SF(lr, at::kFloat, K_AGENT, true), SF(vdes, at::kFloat, K_AGENT, true),
    SF(route_wp, at::kInt, K_AGENT, true), SF(present, at::kByte, K_AGENT, true), SF(collided, at::kByte, K_AGENT, true),
    SF(offroad, at::kByte, K_AGENT, true), SF(scn, at::kInt, K_ENV, true), SF(steps, at::kInt, K_ENV, true),
    SF(target_idx, at::kInt, K_ENV, true), SF(reached, at::kInt, K_ENV, true), SF(episode, at::kInt, K_ENV, true),
    SF(action, at::kFloat, K_ENV2, true), SF(reward, at::kFloat, K_ENV, true), SF(terminated, at::kByte, K_ENV, true),
    SF(truncated, at::kByte, K_ENV, true), SF(tl_violation, at::kByte, K_ENV, true), SF(info, at::kDouble, K_ENV4, false),
    SF(info_reached, at::kInt, K_ENV, false), SF(done_bits, at::kByte, K_ENV, false), SF(obs, at::kFloat, K_ENV8, false),
    SF(ep_return, at::kDouble, K_ENV, false), SF(ep_final, at::kDouble, K_ENV, false), SF(ep_final_len, at::kInt, K_ENV, false),
    SF(slot_cache, at::kInt, K_SLOT8, false), SF(env_cache, at::kInt, K_ENV8, false), SF(act_cache, at::kInt, K_ACT, false),
    SF(magnitudes, at::kFloat, K_ENV4, false)};
#undef SF

int64_t count_of(int kind, int64_t B, int64_t A)
{
    switch (kind) {
        case K_AGENT: return B * A;
        case K_ENV: return B;
        case K_ENV2: return 2 * B;
        case K_ENV4: return 4 * B;
        case K_ENV8: return 8 * B;
        case K_SLOT8: return 8 * B * A;
        case K_ACT: return 2 * B * (A + 1);
        default: return -1;
    }
}

// fills the pointer members of a struct from a dict of named tensors; every tensor used is appended to `keep`
template <size_t N>
void fill_struct(void *st, const Field (&fields)[N], const py::dict &tensors, int64_t B, int64_t A, const at::Device &dev,
                 std::vector<at::Tensor> &keep, const char *what)
{
    for (const Field &f : fields) {
        const void *p = nullptr;
        if (tensors.contains(f.name) && !tensors[f.name].is_none()) {
            const at::Tensor t = py::cast<at::Tensor>(tensors[f.name]);
            p = dev_ptr(t, f.dt, count_of(f.kind, B, A), f.name, dev);
            TORCH_CHECK(t.numel() > 0, what, ".", f.name, " is empty");
            keep.push_back(t);
        } else {
            TORCH_CHECK(!f.required, what, " lacks the tensor '", f.name, "'");
        }
        std::memcpy(static_cast<char *>(st) + f.off, &p, sizeof(p));
    }
    for (const auto &kv : tensors) {
        const std::string k = py::cast<std::string>(kv.first);
        bool known = false;
        for (const Field &f : fields) known = known || k == f.name;
        TORCH_CHECK(known, what, " has no field '", k, "'");
    }
}

class World {
  public:
    World(const py::dict &tensors, const py::dict &ints, int64_t device_index) : dev(at::kCUDA, static_cast<c10::DeviceIndex>(device_index))
    {
        std::memset(&w, 0, sizeof(w));
        fill_struct(&w, kWorldFields, tensors, 0, 0, dev, keep_, "world");
        auto geti = [&](const char *k) {
            TORCH_CHECK(ints.contains(k), "world lacks the size '", k, "'");
            return static_cast<int32_t>(py::cast<int64_t>(ints[k]));
        };
        w.n_maps = geti("n_maps"); w.n_scn = geti("n_scn"); w.NW = geti("NW"); w.A = geti("A");
        w.n_routes = geti("n_routes"); w.RW = geti("RW"); w.n_replay = geti("n_replay"); w.RT = geti("RT");
        w.hints = geti("hints"); w.NH = geti("NH");
        // the tables whose sizes the struct's integers imply
        auto bytes_of = [&](const char *k) { const at::Tensor t = py::cast<at::Tensor>(tensors[k]); return (int64_t)t.numel() * (int64_t)t.element_size(); };
        TORCH_CHECK(w.n_maps >= 1 && w.n_scn >= 1 && w.NW >= 2 && w.A >= 1, "world sizes out of range");
        TORCH_CHECK(bytes_of("maps") == (int64_t)w.n_maps * (int64_t)sizeof(tde_map), "world.maps: expected n_maps * sizeof(tde_map) bytes");
        TORCH_CHECK(bytes_of("scn") == (int64_t)w.n_scn * (int64_t)sizeof(tde_scenario), "world.scn: expected n_scn * sizeof(tde_scenario) bytes");
        TORCH_CHECK(bytes_of("spawn") == (int64_t)w.n_scn * w.A * (int64_t)sizeof(tde_spawn), "world.spawn: expected n_scn * A * sizeof(tde_spawn) bytes");
        TORCH_CHECK(bytes_of("wp_xy") == (int64_t)w.n_scn * w.NW * 16, "world.wp_xy: expected [n_scn][NW][2] float64");
        TORCH_CHECK(bytes_of("route_xy") >= (int64_t)w.n_routes * w.RW * 8, "world.route_xy: smaller than [n_routes][RW][2] float32");
        TORCH_CHECK(bytes_of("replay_states") >= (int64_t)w.n_replay * w.RT * 16, "world.replay_states: smaller than [n_replay][RT][4] float32");
        TORCH_CHECK(w.NH >= 0 && bytes_of("start_psi") >= (int64_t)w.n_scn * w.NH * 4, "world.start_psi: smaller than [n_scn][NH] float32");
        TORCH_CHECK(bytes_of("first_gap") == (int64_t)w.n_scn * w.A * (int64_t)sizeof(tde_first_gap), "world.first_gap: expected n_scn * A * sizeof(tde_first_gap) bytes");
    }
    tde_world w;
    at::Device dev;

  private:
    std::vector<at::Tensor> keep_;
};

class EnvHandle {
  public:
    EnvHandle(const Config &cfg, std::shared_ptr<World> world, const py::dict &state, int64_t B, int64_t A)
        : cfg_(cfg.c), world_(world->w), wkeep_(std::move(world)), dev_(wkeep_->dev)
    {
        TORCH_CHECK(tde_abi_version() == TDE_ABI_VERSION, "libtde_hip.so ABI ", tde_abi_version(), " != header ", TDE_ABI_VERSION);
        TORCH_CHECK(B >= 0 && A >= 1 && A <= TDE_MAX_AGENTS && (A & (A - 1)) == 0, "A must be a power of two <= ", TDE_MAX_AGENTS);
        std::memset(&state_, 0, sizeof(state_));
        fill_struct(&state_, kStateFields, state, B, A, dev_, keep_, "state");
        state_.B = static_cast<int32_t>(B);
        state_.A = static_cast<int32_t>(A);
    }

    // tde_env_step: one timestep of every env; `action` float32 [B, 2] on the device, read in place
    void step(const at::Tensor &action, int64_t flags)
    {
        tde_state st = state_;
        st.action = static_cast<const float *>(dev_ptr(action, at::kFloat, 2 * (int64_t)state_.B, "action", dev_));
        cfg_.flags = static_cast<uint32_t>(flags);
        const c10::hip::HIPGuardMasqueradingAsCUDA guard(dev_);
        check_rc(tde_env_step(&cfg_, &world_, &st, c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev_.index()).stream()), "tde_env_step");
    }

    void reset(const std::optional<at::Tensor> &mask, int64_t flags)
    {
        cfg_.flags = static_cast<uint32_t>(flags);
        const uint8_t *m = mask ? static_cast<const uint8_t *>(dev_ptr(*mask, at::kByte, state_.B, "mask", dev_)) : nullptr;
        const c10::hip::HIPGuardMasqueradingAsCUDA guard(dev_);
        check_rc(tde_env_reset(&cfg_, &world_, &state_, m, c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev_.index()).stream()), "tde_env_reset");
    }

    void rollout(const at::Tensor &actions, const at::Tensor &reward, const at::Tensor &done, int64_t flags)
    {
        cfg_.flags = static_cast<uint32_t>(flags);
        TORCH_CHECK(actions.dim() == 3 && actions.size(1) == state_.B && actions.size(2) == 2, "actions must be [K, B, 2]");
        const int64_t K = actions.size(0);
        tde_rollout ro;
        ro.actions = static_cast<const float *>(dev_ptr(actions, at::kFloat, K * state_.B * 2, "actions", dev_));
        ro.reward = static_cast<float *>(const_cast<void *>(dev_ptr(reward, at::kFloat, K * state_.B, "reward", dev_)));
        ro.done = static_cast<uint8_t *>(const_cast<void *>(dev_ptr(done, at::kByte, K * state_.B, "done", dev_)));
        ro.K = static_cast<int32_t>(K);
        ro.ldb = 0;
        const c10::hip::HIPGuardMasqueradingAsCUDA guard(dev_);
        check_rc(tde_env_rollout(&cfg_, &world_, &state_, &ro, c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev_.index()).stream()), "tde_env_rollout");
    }

    // tde_env_step_render: the timestep + the birdview as streams.size() sub-batches, each on its own stream (raw hipStream_t
    // values, e.g. torch.cuda.Stream.cuda_stream); `out` = None: the step only
    void step_render(const at::Tensor &action, int64_t flags, const std::optional<at::Tensor> &out, int64_t H, int64_t W, double fov,
                     int64_t n_stack, const std::optional<at::Tensor> &layers, int64_t phase, int64_t rflags,
                     const std::optional<at::Tensor> &fresh, const std::vector<int64_t> &streams)
    {
        tde_state st = state_;
        st.action = static_cast<const float *>(dev_ptr(action, at::kFloat, 2 * (int64_t)state_.B, "action", dev_));
        cfg_.flags = static_cast<uint32_t>(flags);
        const int64_t ns = n_stack > 1 ? n_stack : 1;
        tde_render rd{};
        if (out) {
            rd.out = static_cast<uint8_t *>(const_cast<void *>(dev_ptr(*out, at::kByte, state_.B * 3 * ns * H * W, "out", dev_)));
            rd.H = static_cast<int32_t>(H);
            rd.W = static_cast<int32_t>(W);
            rd.fov = static_cast<float>(fov);
            rd.n_stack = static_cast<int32_t>(n_stack);
            rd.layers = layers ? static_cast<uint8_t *>(const_cast<void *>(dev_ptr(*layers, at::kByte, state_.B * ns * H * W, "layers", dev_))) : nullptr;
            rd.phase = static_cast<int32_t>(phase);
            rd.flags = static_cast<int32_t>(rflags);
            rd.fresh = fresh ? static_cast<const uint8_t *>(dev_ptr(*fresh, at::kByte, state_.B, "fresh", dev_)) : nullptr;
            rd.only = nullptr;
        }
        TORCH_CHECK(!streams.empty() && streams.size() <= 16, "streams: 1 to 16 raw stream handles");
        void *sv[16];
        for (size_t i = 0; i < streams.size(); ++i) sv[i] = reinterpret_cast<void *>(static_cast<uintptr_t>(streams[i]));
        const c10::hip::HIPGuardMasqueradingAsCUDA guard(dev_);
        check_rc(tde_env_step_render(&cfg_, &world_, &st, out ? &rd : nullptr, sv, static_cast<int32_t>(streams.size())), "tde_env_step_render");
    }

    void render(const at::Tensor &out, int64_t H, int64_t W, double fov, int64_t n_stack, const std::optional<at::Tensor> &layers,
                int64_t phase, int64_t flags, const std::optional<at::Tensor> &fresh, const std::optional<at::Tensor> &only)
    {
        const int64_t ns = n_stack > 1 ? n_stack : 1;
        tde_render rd;
        rd.out = static_cast<uint8_t *>(const_cast<void *>(dev_ptr(out, at::kByte, state_.B * 3 * ns * H * W, "out", dev_)));
        rd.H = static_cast<int32_t>(H);
        rd.W = static_cast<int32_t>(W);
        rd.fov = static_cast<float>(fov);
        rd.n_stack = static_cast<int32_t>(n_stack);
        rd.layers = layers ? static_cast<uint8_t *>(const_cast<void *>(dev_ptr(*layers, at::kByte, state_.B * ns * H * W, "layers", dev_))) : nullptr;
        rd.phase = static_cast<int32_t>(phase);
        rd.flags = static_cast<int32_t>(flags);
        rd.fresh = fresh ? static_cast<const uint8_t *>(dev_ptr(*fresh, at::kByte, state_.B, "fresh", dev_)) : nullptr;
        rd.only = only ? static_cast<const uint8_t *>(dev_ptr(*only, at::kByte, state_.B, "only", dev_)) : nullptr;
        const c10::hip::HIPGuardMasqueradingAsCUDA guard(dev_);
        check_rc(tde_render_ego(&cfg_, &world_, &state_, &rd, c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev_.index()).stream()), "tde_render_ego");
    }

    // tde_env_reset_render: masked reset + the re-spawned views' first observation (their newest frame in place, older stack frames
    // blank) in one call; `phase` = the phase of the last full render
    void reset_render(const at::Tensor &mask, int64_t flags, const at::Tensor &out, int64_t H, int64_t W, double fov, int64_t n_stack,
                      const std::optional<at::Tensor> &layers, int64_t phase, int64_t rflags)
    {
        cfg_.flags = static_cast<uint32_t>(flags);
        const int64_t ns = n_stack > 1 ? n_stack : 1;
        tde_render rd{};
        rd.out = static_cast<uint8_t *>(const_cast<void *>(dev_ptr(out, at::kByte, state_.B * 3 * ns * H * W, "out", dev_)));
        rd.H = static_cast<int32_t>(H); rd.W = static_cast<int32_t>(W); rd.fov = static_cast<float>(fov);
        rd.n_stack = static_cast<int32_t>(n_stack);
        rd.layers = layers ? static_cast<uint8_t *>(const_cast<void *>(dev_ptr(*layers, at::kByte, state_.B * ns * H * W, "layers", dev_))) : nullptr;
        rd.phase = static_cast<int32_t>(phase);
        rd.flags = static_cast<int32_t>(rflags);
        const uint8_t *m = static_cast<const uint8_t *>(dev_ptr(mask, at::kByte, state_.B, "mask", dev_));
        const c10::hip::HIPGuardMasqueradingAsCUDA guard(dev_);
        check_rc(tde_env_reset_render(&cfg_, &world_, &state_, m, &rd, c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev_.index()).stream()), "tde_env_reset_render");
    }

    void state_obs(const at::Tensor &out)
    {
        float *p = static_cast<float *>(const_cast<void *>(dev_ptr(out, at::kFloat, (int64_t)state_.B * 8, "out", dev_)));
        const c10::hip::HIPGuardMasqueradingAsCUDA guard(dev_);
        check_rc(tde_state_obs(&world_, &state_, p, c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev_.index()).stream()), "tde_state_obs");
    }

    // tde_ego_infractions: float32 [B, 4] = the ego's (offroad, collision, overlap count, 0) magnitudes of the state as it is (gym_env.py:427-428)
    void ego_infractions(const at::Tensor &out, int64_t flags)
    {
        float *p = static_cast<float *>(const_cast<void *>(dev_ptr(out, at::kFloat, (int64_t)state_.B * 4, "out", dev_)));
        cfg_.flags = static_cast<uint32_t>(flags);
        const c10::hip::HIPGuardMasqueradingAsCUDA guard(dev_);
        check_rc(tde_ego_infractions(&cfg_, &world_, &state_, p, c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev_.index()).stream()), "tde_ego_infractions");
    }

    // tde_env_post_step: magnitudes (optional) of the state a step without TDE_F_AUTORESET left + the re-spawn of the envs it finished
    void post_step(const std::optional<at::Tensor> &mag, int64_t flags)
    {
        float *p = mag ? static_cast<float *>(const_cast<void *>(dev_ptr(*mag, at::kFloat, (int64_t)state_.B * 4, "magnitudes", dev_))) : nullptr;
        cfg_.flags = static_cast<uint32_t>(flags);
        const c10::hip::HIPGuardMasqueradingAsCUDA guard(dev_);
        check_rc(tde_env_post_step(&cfg_, &world_, &state_, p, c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev_.index()).stream()), "tde_env_post_step");
    }

    int64_t flags() const { return cfg_.flags; }
    int64_t num_envs() const { return state_.B; }
    int64_t agents_per_env() const { return state_.A; }

  private:
    tde_config cfg_;
    tde_world world_;
    std::shared_ptr<World> wkeep_;         // (owns the world's tensors)
    tde_state state_;
    at::Device dev_;
    std::vector<at::Tensor> keep_;         // the state's tensors
};

// ---- operator level: the SimulatorInterface methods GymEnv calls (include/tde_hip.h, first block), torch tensors in / out ----
template <typename T> T *ptr(const at::Tensor &t, at::ScalarType dt, int64_t numel, const char *name, const at::Device &dev)
{
    return static_cast<T *>(const_cast<void *>(dev_ptr(t, dt, numel, name, dev)));
}
void *cur_stream(const at::Device &dev) { return c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev.index()).stream(); }

// tde_kinematics_step: KinematicBicycle.step for n agents, in place (ref gym_env.py:117)
void kinematics_step(const at::Tensor &x, const at::Tensor &y, const at::Tensor &psi, const at::Tensor &v, const at::Tensor &lr,
                     const at::Tensor &action, const std::optional<at::Tensor> &present, double dt)
{
    const int64_t n = x.numel();
    const at::Device dev = x.device();
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);
    check_rc(tde_kinematics_step(n, ptr<float>(x, at::kFloat, n, "x", dev), ptr<float>(y, at::kFloat, n, "y", dev),
                                 ptr<float>(psi, at::kFloat, n, "psi", dev), ptr<float>(v, at::kFloat, n, "v", dev),
                                 ptr<float>(lr, at::kFloat, n, "lr", dev),
                                 present ? ptr<uint8_t>(*present, at::kByte, n, "present", dev) : nullptr,
                                 ptr<float>(action, at::kFloat, 2 * n, "action", dev), (float)dt, cur_stream(dev)),
             "tde_kinematics_step");
}

// tde_compute_collision: compute_collision() > 0 per agent -> uint8 [B * A] (ref gym_env.py:143)
at::Tensor compute_collision(int64_t B, int64_t A, const at::Tensor &x, const at::Tensor &y, const at::Tensor &psi,
                             const at::Tensor &length, const at::Tensor &width, const at::Tensor &present)
{
    const int64_t n = B * A;
    const at::Device dev = x.device();
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);
    at::Tensor out = at::empty({n}, at::TensorOptions().dtype(at::kByte).device(dev));
    check_rc(tde_compute_collision((int32_t)B, (int32_t)A, ptr<float>(x, at::kFloat, n, "x", dev), ptr<float>(y, at::kFloat, n, "y", dev),
                                   ptr<float>(psi, at::kFloat, n, "psi", dev), ptr<float>(length, at::kFloat, n, "length", dev),
                                   ptr<float>(width, at::kFloat, n, "width", dev), ptr<uint8_t>(present, at::kByte, n, "present", dev),
                                   out.data_ptr<uint8_t>(), cur_stream(dev)),
             "tde_compute_collision");
    return out;
}

// tde_compute_offroad: compute_offroad() > 0 per agent -> uint8 [B * A] (ref gym_env.py:142)
at::Tensor compute_offroad(int64_t B, int64_t A, const at::Tensor &x, const at::Tensor &y, const at::Tensor &psi,
                           const at::Tensor &length, const at::Tensor &width, const at::Tensor &present, const World &world,
                           const at::Tensor &map_of_env, double threshold)
{
    const int64_t n = B * A;
    const at::Device dev = x.device();
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);
    at::Tensor out = at::empty({n}, at::TensorOptions().dtype(at::kByte).device(dev));
    check_rc(tde_compute_offroad((int32_t)B, (int32_t)A, ptr<float>(x, at::kFloat, n, "x", dev), ptr<float>(y, at::kFloat, n, "y", dev),
                                 ptr<float>(psi, at::kFloat, n, "psi", dev), ptr<float>(length, at::kFloat, n, "length", dev),
                                 ptr<float>(width, at::kFloat, n, "width", dev), ptr<uint8_t>(present, at::kByte, n, "present", dev),
                                 &world.w, ptr<int32_t>(map_of_env, at::kInt, B, "map_of_env", dev),
                                 (float)threshold, out.data_ptr<uint8_t>(), cur_stream(dev)),
             "tde_compute_offroad");
    return out;
}

// tde_waypoint_reward (ref gym_env.py:391-437): pre / post = (x, y, psi, v) stacked [4][n]; steps / target_idx / reached
// int32 [n], updated in place; -> (reward f32 [n], terminated u8 [n], truncated u8 [n], info f64 [n][4], info_reached i32 [n])
std::vector<at::Tensor> waypoint_reward(const Config &cfg, const at::Tensor &pre, const at::Tensor &post, const at::Tensor &offroad,
                                        const at::Tensor &collided, const std::optional<at::Tensor> &tl, const at::Tensor &wp_xy,
                                        const at::Tensor &wp_n, const at::Tensor &scn, const at::Tensor &steps,
                                        const at::Tensor &target_idx, const at::Tensor &reached)
{
    TORCH_CHECK(pre.dim() == 2 && pre.size(0) == 4 && post.dim() == 2 && post.size(0) == 4, "pre / post must be [4][n]");
    TORCH_CHECK(wp_xy.dim() == 3 && wp_xy.size(2) == 2, "wp_xy must be [S][NW][2]");
    const int64_t n = pre.size(1), S = wp_xy.size(0), NW = wp_xy.size(1);
    const at::Device dev = pre.device();
    const c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);
    const float *p0 = ptr<float>(pre, at::kFloat, 4 * n, "pre", dev), *p1 = ptr<float>(post, at::kFloat, 4 * n, "post", dev);
    const auto opt = at::TensorOptions().device(dev);
    at::Tensor reward = at::empty({n}, opt.dtype(at::kFloat)), term = at::empty({n}, opt.dtype(at::kByte)),
               trunc = at::empty({n}, opt.dtype(at::kByte)), info = at::empty({n, 4}, opt.dtype(at::kDouble)),
               info_reached = at::empty({n}, opt.dtype(at::kInt));
    check_rc(tde_waypoint_reward(&cfg.c, (int32_t)n, p0, p0 + n, p0 + 2 * n, p0 + 3 * n, p1,
                                 p1 + n, p1 + 2 * n, p1 + 3 * n, ptr<uint8_t>(offroad, at::kByte, n, "offroad", dev),
                                 ptr<uint8_t>(collided, at::kByte, n, "collided", dev),
                                 tl ? ptr<uint8_t>(*tl, at::kByte, n, "tl", dev) : nullptr,
                                 ptr<double>(wp_xy, at::kDouble, S * NW * 2, "wp_xy", dev), ptr<int32_t>(wp_n, at::kInt, S, "wp_n", dev),
                                 (int32_t)NW, ptr<int32_t>(scn, at::kInt, n, "scn", dev), ptr<int32_t>(steps, at::kInt, n, "steps", dev),
                                 ptr<int32_t>(target_idx, at::kInt, n, "target_idx", dev), ptr<int32_t>(reached, at::kInt, n, "reached", dev),
                                 reward.data_ptr<float>(), term.data_ptr<uint8_t>(), trunc.data_ptr<uint8_t>(), info.data_ptr<double>(),
                                 info_reached.data_ptr<int32_t>(), cur_stream(dev)),
             "tde_waypoint_reward");
    return {reward, term, trunc, info, info_reached};
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.doc() = "PyTorch-ROCm C++ extension over the C-ABI of libtde_hip.so (include/tde_hip.h)";
    m.def("abi_version", []() { return tde_abi_version(); });
    m.def("kinematics_step", &kinematics_step, py::arg("x"), py::arg("y"), py::arg("psi"), py::arg("v"), py::arg("lr"),
          py::arg("action"), py::arg("present") = py::none(), py::arg("dt") = 0.1);
    m.def("compute_collision", &compute_collision, py::arg("B"), py::arg("A"), py::arg("x"), py::arg("y"), py::arg("psi"),
          py::arg("length"), py::arg("width"), py::arg("present"));
    m.def("compute_offroad", &compute_offroad, py::arg("B"), py::arg("A"), py::arg("x"), py::arg("y"), py::arg("psi"),
          py::arg("length"), py::arg("width"), py::arg("present"), py::arg("world"), py::arg("map_of_env"),
          py::arg("threshold") = 0.5);
    m.def("waypoint_reward", &waypoint_reward, py::arg("cfg"), py::arg("pre"), py::arg("post"), py::arg("offroad"),
          py::arg("collided"), py::arg("tl"), py::arg("wp_xy"), py::arg("wp_n"), py::arg("scn"), py::arg("steps"),
          py::arg("target_idx"), py::arg("reached"));
    py::class_<Config>(m, "Config", "tde_config by value, from its bytes").def(py::init<const py::bytes &>(), py::arg("raw"));
    py::class_<World, std::shared_ptr<World>>(m, "World", "tde_world over named device tensors (kept alive)")
        .def(py::init<const py::dict &, const py::dict &, int64_t>(), py::arg("tensors"), py::arg("ints"), py::arg("device_index"));
    py::class_<EnvHandle>(m, "EnvHandle")
        .def(py::init<const Config &, std::shared_ptr<World>, const py::dict &, int64_t, int64_t>(), py::arg("cfg"), py::arg("world"),
             py::arg("state"), py::arg("B"), py::arg("A"))
        .def("step", &EnvHandle::step, py::arg("action"), py::arg("flags"))
        .def("reset", &EnvHandle::reset, py::arg("mask"), py::arg("flags"))
        .def("rollout", &EnvHandle::rollout, py::arg("actions"), py::arg("reward"), py::arg("done"), py::arg("flags"))
        .def("render", &EnvHandle::render, py::arg("out"), py::arg("H"), py::arg("W"), py::arg("fov"), py::arg("n_stack"),
             py::arg("layers"), py::arg("phase"), py::arg("flags"), py::arg("fresh"), py::arg("only"))
        .def("step_render", &EnvHandle::step_render, py::arg("action"), py::arg("flags"), py::arg("out"), py::arg("H"), py::arg("W"),
             py::arg("fov"), py::arg("n_stack"), py::arg("layers"), py::arg("phase"), py::arg("rflags"), py::arg("fresh"), py::arg("streams"))
        .def("reset_render", &EnvHandle::reset_render, py::arg("mask"), py::arg("flags"), py::arg("out"), py::arg("H"), py::arg("W"), py::arg("fov"),
             py::arg("n_stack"), py::arg("layers"), py::arg("phase"), py::arg("rflags"))
        .def("state_obs", &EnvHandle::state_obs)
        .def("ego_infractions", &EnvHandle::ego_infractions, py::arg("out"), py::arg("flags"))
        .def("post_step", &EnvHandle::post_step, py::arg("magnitudes"), py::arg("flags"))
        .def_property_readonly("flags", &EnvHandle::flags)
        .def_property_readonly("num_envs", &EnvHandle::num_envs)
        .def_property_readonly("agents_per_env", &EnvHandle::agents_per_env);
}
