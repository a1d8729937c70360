// vec_env.cpp - native host loop for batches of CircuitEnv environments (C ABI: include/vqe_env.h).
//
// Restates, on sparse per-environment gate lists, the bookkeeping the reference's CircuitEnv.step()
// performs in Python on a dense (L, n+6, n) float32 tensor
// (environments/environment_qulacs_TN_notin_agent.py:230-333, :484-627; environment_qulacs.py:169-267):
//   * a gate list kept in construct_ansatz order - per layer the CNOTs by (target, control), then the
//     rotations by (axis, qubit) (environments/VQAs/VQE_qulacs_TN_notin_RL.py:13-45) - replaces the tensor;
//     parameter j is the j-th rotation of the list, as in scipy_optim (:454-456);
//   * one vqe_batch_load + vqe_batch_run_env_step (include/vqe_hip.h) serves all environments of the batch.
// Host code only (no device code in this translation unit); linked into libvqe_hip.so.
#include "../../include/vqe_env.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <new>
#include <string>
#include <vector>

namespace {

struct Gate {
  int32_t layer;
  int8_t kind;      // 0 CNOT, 1..3 RX / RY / RZ
  int8_t q0, q1;    // CNOT: control, target; rotation: qubit, -1
  float angle;
  // position in construct_ansatz order
  uint32_t key() const {
    const uint32_t a = kind == 0 ? (uint32_t)q1 : (uint32_t)(kind - 1);   // CNOT: target row; rotation: axis row
    const uint32_t b = (uint32_t)q0;                                      // CNOT: control column; rotation: qubit column
    return ((uint32_t)layer << 14) | ((kind != 0 ? 1u : 0u) << 13) | (a << 6) | b;
  }
};

struct Action { int32_t v[4]; };
inline bool operator==(const Action& a, const Action& b) { return std::memcmp(a.v, b.v, sizeof a.v) == 0; }

struct Env {
  std::vector<Gate> gates;
  std::vector<int32_t> moments;
  std::vector<Action> slot;        // illegal-action slots
  std::vector<uint8_t> slot_used;
  Action current{}, previous{};
  int32_t step_counter = -1, halting_step = -1, nfev = 0, new_pos = -1, new_param = -1;
  double prev_energy = 0, energy = 0, error = 0, rwd = 0, done_threshold = 0;
  // VanillaCurriculum: `curriculum` is the episode's copy, `curriculum_saved` what reset() copies from
  int64_t episodes_completed = 0, episodes_completed_saved = 0;
  double lowest_energy = 0, lowest_energy_saved = 0;
  std::vector<double> opt_ang;
  int64_t obs_index = -1;
};

}  // namespace

struct vqe_vecenv {
  vqe_vecenv_config_t cfg{};
  vqe_t* eng = nullptr;
  std::vector<double> thresholds;
  std::vector<int64_t> switch_episodes;
  std::vector<Gate> init_gates;
  std::map<std::vector<int32_t>, int32_t> action_index;
  std::vector<Env> env;
  std::string err;
  bool pending = false;
  // flat batch description (reused)
  std::vector<int64_t> gate_off, par_off;
  std::vector<int32_t> kind, q0, q1, pidx, new_gate, nfev;
  std::vector<double> theta, x, xraw, f;
};

namespace {

int fail(vqe_vecenv* v, int code, const std::string& msg) {
  if (v) v->err = msg;
  return code;
}

double current_threshold(const vqe_vecenv* v, int64_t episodes_completed, bool* ok) {
  // thresholds[min{i : switch_episodes[i] > episodes_completed}] (curricula.py:93-96)
  for (size_t i = 0; i < v->switch_episodes.size(); ++i)
    if (v->switch_episodes[i] > episodes_completed) { *ok = true; return v->thresholds[i]; }
  *ok = false;
  return 0.0;
}

// CircuitEnv.illegal_action_new (environment_qulacs_TN_notin_agent.py:502-627), slot for slot
void illegal_update(const vqe_vecenv* v, Env& e) {
  const int n = v->cfg.n_qubits;
  const Action act = e.current;
  const int ctrl = act.v[0], targ = (act.v[0] + act.v[1]) % n, rot_qubit = act.v[2], rot_axis = act.v[3];
  auto park = [&]() {                     // first free slot among 1..n-1 takes the action
    for (int i = 1; i < n; ++i)
      if (!e.slot_used[i]) { e.slot[i] = act; e.slot_used[i] = 1; return; }
  };
  auto occupied = [&]() {
    for (int i = 0; i < n; ++i) if (e.slot_used[i]) return true;
    return false;
  };
  if (ctrl < n) {
    if (occupied()) {
      for (int k = 0; k < n; ++k) {       // live iteration: sees the slots changed below
        if (!e.slot_used[k]) continue;
        const Action old = e.slot[k];
        const int old_targ = (old.v[0] + old.v[1]) % n;
        bool clash;
        if (old.v[2] == n) clash = ctrl == old.v[0] || ctrl == old_targ || targ == old.v[0] || targ == old_targ;
        else clash = old.v[2] == ctrl || old.v[2] == targ;
        if (clash) e.slot_used[k] = 0;
        park();
      }
    } else {
      e.slot[0] = act; e.slot_used[0] = 1;
    }
  }
  if (rot_qubit < n) {
    if (occupied()) {
      for (int k = 0; k < n; ++k) {
        if (!e.slot_used[k]) continue;
        const Action old = e.slot[k];
        const int old_targ = (old.v[0] + old.v[1]) % n;
        if (old.v[0] == n) {
          if (rot_qubit == old.v[2]) {
            if (rot_axis != old.v[3]) { e.slot_used[k] = 0; park(); }
          } else {
            park();
          }
        } else {
          if (rot_qubit == old.v[0] || rot_qubit == old_targ) e.slot_used[k] = 0;
          park();
        }
      }
    } else {
      e.slot[0] = act; e.slot_used[0] = 1;
    }
  }
  auto same = [&](int i, int j) {          // Python list equality: two empty slots are equal
    if (!e.slot_used[i] || !e.slot_used[j]) return !e.slot_used[i] && !e.slot_used[j];
    return e.slot[i] == e.slot[j];
  };
  for (int i = 0; i < n; ++i)
    for (int j = i + 1; j < n; ++j)
      if (same(i, j)) {
        if (j != i + 1) e.slot_used[i] = 0; else e.slot_used[j] = 0;
        break;
      }
  for (int i = 0; i + 1 < n; ++i)
    if (!e.slot_used[i]) {
      e.slot[i] = e.slot[i + 1]; e.slot_used[i] = e.slot_used[i + 1];
      e.slot_used[i + 1] = 0;
    }
}

void reset_env(vqe_vecenv* v, Env& e, int32_t halting) {
  const int n = v->cfg.n_qubits;
  e.gates = v->init_gates;
  e.moments.assign(n, 0);
  e.slot.assign(n, Action{});
  e.slot_used.assign(n, 0);
  for (int k = 0; k < 4; ++k) e.current.v[k] = n;
  e.step_counter = -1;
  e.halting_step = halting;
  e.episodes_completed = e.episodes_completed_saved;      // curriculum = deepcopy(curriculum_dict[prob]) (:366)
  e.lowest_energy = e.lowest_energy_saved;
  bool ok;
  e.done_threshold = current_threshold(v, e.episodes_completed, &ok);
  e.prev_energy = v->cfg.init_energy;
  e.new_pos = e.new_param = -1;
  e.obs_index = -1;
}

}  // namespace

extern "C" {

int vqe_vecenv_create(const vqe_vecenv_config_t* c, vqe_t* engine, vqe_vecenv_t** out) {
  if (!out) return VQE_EINVAL;
  *out = nullptr;
  if (!c || !engine || c->n_qubits < 1 || c->n_qubits > 30 || c->num_layers < 1 || c->num_envs < 1 || c->maxfun < 1 ||
      c->n_thresholds < 1 || !c->thresholds || !c->switch_episodes || c->n_actions < 1 || !c->action_table ||
      (c->n_init_gates > 0 && (!c->init_layer || !c->init_kind || !c->init_q0 || !c->init_q1 || !c->init_angle)))
    return VQE_EINVAL;
  vqe_vecenv* v = new (std::nothrow) vqe_vecenv;
  if (!v) return VQE_ENOMEM;
  v->cfg = *c;
  v->eng = engine;
  v->thresholds.assign(c->thresholds, c->thresholds + c->n_thresholds);
  v->switch_episodes.assign(c->switch_episodes, c->switch_episodes + c->n_thresholds);
  v->cfg.thresholds = nullptr; v->cfg.switch_episodes = nullptr;
  for (int i = 0; i < c->n_init_gates; ++i) {
    const int k = c->init_kind[i];
    if (k < 0 || k > 3 || c->init_layer[i] < 0 || c->init_layer[i] >= c->num_layers || c->init_q0[i] < 0 || c->init_q0[i] >= c->n_qubits ||
        (k == 0 && (c->init_q1[i] < 0 || c->init_q1[i] >= c->n_qubits))) { delete v; return VQE_EINVAL; }
    v->init_gates.push_back(Gate{c->init_layer[i], (int8_t)k, (int8_t)c->init_q0[i], (int8_t)(k == 0 ? c->init_q1[i] : -1), c->init_angle[i]});
  }
  std::sort(v->init_gates.begin(), v->init_gates.end(), [](const Gate& a, const Gate& b) { return a.key() < b.key(); });
  for (int i = 0; i < c->n_actions; ++i)
    v->action_index[std::vector<int32_t>(c->action_table + 4 * i, c->action_table + 4 * i + 4)] = i;
  v->cfg.action_table = nullptr;
  v->cfg.init_layer = v->cfg.init_kind = v->cfg.init_q0 = v->cfg.init_q1 = nullptr; v->cfg.init_angle = nullptr;
  bool ok;
  (void)current_threshold(v, 0, &ok);
  if (!ok) { delete v; return VQE_EINVAL; }
  v->env.resize(c->num_envs);
  for (auto& e : v->env) {
    e.lowest_energy_saved = c->min_eig + c->accept_err;     // VanillaCurriculum.__init__ (curricula.py:88-91)
    e.episodes_completed_saved = 0;
    reset_env(v, e, -1);
  }
  *out = v;
  return VQE_OK;
}

void vqe_vecenv_destroy(vqe_vecenv_t* v) { delete v; }
const char* vqe_vecenv_last_error(const vqe_vecenv_t* v) { return v ? v->err.c_str() : "invalid arguments"; }

int vqe_vecenv_reset(vqe_vecenv_t* v, int32_t count, const int32_t* idx, const int32_t* halting_step) {
  if (!v) return VQE_EINVAL;
  if (v->pending) return fail(v, VQE_ESTATE, "reset between step_begin and step_end");
  const int B = v->cfg.num_envs;
  if (!idx) count = B;
  for (int i = 0; i < count; ++i) {
    const int b = idx ? idx[i] : i;
    if (b < 0 || b >= B) return fail(v, VQE_EINVAL, "environment index out of range");
    reset_env(v, v->env[b], halting_step ? halting_step[i] : -1);
  }
  return VQE_OK;
}

int vqe_vecenv_illegal_actions(vqe_vecenv_t* v, int32_t* out) {
  if (!v || !out) return VQE_EINVAL;
  const int n = v->cfg.n_qubits, B = v->cfg.num_envs;
  std::vector<int32_t> key(4);
  for (int b = 0; b < B; ++b) {
    Env& e = v->env[b];
    illegal_update(v, e);
    int32_t* o = out + (size_t)b * n;
    int cnt = 0;
    for (int i = 0; i < n; ++i) {
      if (!e.slot_used[i]) continue;
      key.assign(e.slot[i].v, e.slot[i].v + 4);
      auto it = v->action_index.find(key);
      if (it != v->action_index.end()) o[cnt++] = it->second;
    }
    std::sort(o, o + cnt);
    for (int i = cnt; i < n; ++i) o[i] = -1;
  }
  return VQE_OK;
}

int vqe_vecenv_step_begin(vqe_vecenv_t* v, const int32_t* actions) {
  if (!v || !actions) return VQE_EINVAL;
  if (v->pending) return fail(v, VQE_ESTATE, "step_begin called twice without step_end");
  const int n = v->cfg.n_qubits, B = v->cfg.num_envs, L = v->cfg.num_layers, rec = v->cfg.noisy ? 2 : 1;
  v->gate_off.assign(1, 0); v->par_off.assign(1, 0);
  v->kind.clear(); v->q0.clear(); v->q1.clear(); v->pidx.clear(); v->theta.clear();
  v->new_gate.assign(B, -1);
  // pass 1: every action of the batch is validated before ANY environment is touched - a rejected call leaves all B
  // environments exactly as they were (a caller that catches the error may correct the action and step again)
  for (int b = 0; b < B; ++b) {
    const Env& e = v->env[b];
    const int32_t* a = actions + 4 * (size_t)b;
    const int ctrl = a[0], rot_qubit = a[2], rot_axis = a[3];
    if (ctrl < 0 || ctrl > n || a[1] < 0 || rot_qubit < 0 || rot_qubit > n || (rot_qubit < n && (rot_axis < 1 || rot_axis > 3)) ||
        (ctrl >= n && rot_qubit >= n))
      return fail(v, VQE_EINVAL, "action places no gate / is out of range");
    const int targ = (ctrl + a[1]) % n;
    const int gate_tensor = rot_qubit < n ? e.moments[rot_qubit] : std::max(e.moments[ctrl], e.moments[targ]);
    if (v->cfg.layer_offset + gate_tensor >= L) return fail(v, VQE_EINVAL, "action beyond the last layer of the state tensor");
    if (ctrl < n && targ == ctrl) return fail(v, VQE_EINVAL, "CNOT with control == target");
  }
  // pass 2: the bookkeeping of step() (no failure path from here to the engine calls)
  for (int b = 0; b < B; ++b) {
    Env& e = v->env[b];
    const int32_t* a = actions + 4 * (size_t)b;
    const int ctrl = a[0], rot_qubit = a[2], rot_axis = a[3];
    const int targ = (ctrl + a[1]) % n;
    e.step_counter += 1;                                                     // (:241)
    const bool is_rot = rot_qubit < n;                                       // the layer follows the rotation when both are set (:260-263)
    const int gate_tensor = is_rot ? e.moments[rot_qubit] : std::max(e.moments[ctrl], e.moments[targ]);
    const int layer = v->cfg.layer_offset + gate_tensor;
    // what the action writes: a CNOT when ctrl < n, else the rotation (:265-268)
    Gate g{layer, (int8_t)(ctrl < n ? 0 : rot_axis), (int8_t)(ctrl < n ? ctrl : rot_qubit), (int8_t)(ctrl < n ? targ : -1), 0.0f};
    auto it = std::lower_bound(e.gates.begin(), e.gates.end(), g, [](const Gate& x, const Gate& y) { return x.key() < y.key(); });
    e.new_pos = -1;
    e.obs_index = -1;
    if (it == e.gates.end() || it->key() != g.key()) {      // an occupied slot stays as it is: nothing new to skip
      e.new_pos = (int32_t)(it - e.gates.begin());
      e.gates.insert(it, g);
      const int row = g.kind == 0 ? g.q1 : n + g.kind - 1, col = g.q0;
      e.obs_index = ((int64_t)layer * (n + 3) + row) * n + col;
    }
    if (is_rot) e.moments[rot_qubit] += 1;                                   // (:270-275)
    else { const int m = std::max(e.moments[ctrl], e.moments[targ]); e.moments[ctrl] = e.moments[targ] = m + 1; }
    std::memcpy(e.current.v, a, sizeof e.current.v);
    illegal_update(v, e);                                                    // (:277-278)
    // circuit of the post-action state; x0 = committed float32 angles, the new rotation enters with 0
    int np = 0;
    e.new_param = -1;
    for (size_t i = 0; i < e.gates.size(); ++i) {
      const Gate& q = e.gates[i];
      v->kind.push_back(q.kind); v->q0.push_back(q.q0); v->q1.push_back(q.kind == 0 ? q.q1 : -1);
      if (q.kind != 0) {
        if ((int32_t)i == e.new_pos) e.new_param = np;
        v->pidx.push_back(np++);
        v->theta.push_back((double)q.angle);
      } else {
        v->pidx.push_back(-1);
      }
      if (rec == 2) {
        v->kind.push_back(q.kind == 0 ? VQE_GATE_DEPOL2 : VQE_GATE_DEPOL1);
        v->q0.push_back(q.q0); v->q1.push_back(q.kind == 0 ? q.q1 : -1); v->pidx.push_back(-1);
      }
    }
    v->new_gate[b] = e.new_pos >= 0 ? e.new_pos * rec : -1;
    v->gate_off.push_back((int64_t)v->kind.size());
    v->par_off.push_back((int64_t)v->theta.size());
  }
  int rc = vqe_batch_load(v->eng, B, v->gate_off.data(), v->kind.data(), v->q0.data(), v->q1.data(), v->pidx.data(),
                          v->par_off.data(), v->theta.data());
  if (!rc) rc = vqe_batch_set_new_gate(v->eng, v->new_gate.data());
  if (!rc) rc = vqe_batch_run_env_step(v->eng, 1.0, 1e-4, v->cfg.maxfun);   // scipy 1.15 COBYLA defaults (:478)
  if (rc) return fail(v, rc, std::string("engine: ") + vqe_last_error(v->eng));
  v->pending = true;
  return VQE_OK;
}

int vqe_vecenv_step_end(vqe_vecenv_t* v, int train_flag, float* reward, int32_t* done, int64_t* obs_index) {
  if (!v || !reward || !done) return VQE_EINVAL;
  if (!v->pending) return fail(v, VQE_ESTATE, "step_end without step_begin");
  v->pending = false;
  const int B = v->cfg.num_envs;
  const size_t PT = (size_t)v->par_off.back();
  v->x.resize(PT + 1); v->xraw.resize(PT + 1); v->f.resize(B); v->nfev.resize(B);
  int rc = vqe_batch_fetch(v->eng, v->x.data(), v->f.data(), v->nfev.data());
  if (!rc) rc = vqe_batch_fetch_xopt(v->eng, v->xraw.data());
  if (rc) return fail(v, rc, std::string("engine: ") + vqe_last_error(v->eng));
  bool curriculum_exhausted = false;
  for (int b = 0; b < B; ++b) {
    Env& e = v->env[b];
    const double* xb = v->x.data() + v->par_off[b];
    const double* xr = v->xraw.data() + v->par_off[b];
    const int P = (int)(v->par_off[b + 1] - v->par_off[b]);
    int j = 0;
    for (auto& g : e.gates) if (g.kind != 0) g.angle = (float)xb[j++];        // thetas as torch.float (:285-287, :480)
    e.opt_ang.clear();                                                         // result.x: the pre-action parameters only
    for (int k = 0; k < P; ++k) if (k != e.new_param) e.opt_ang.push_back(xr[k]);
    const double energy = v->f[b];
    e.energy = energy;
    if (energy < e.lowest_energy && train_flag) e.lowest_energy = energy;     // (:299-300)
    e.error = std::fabs(v->cfg.min_eig - energy);
    const bool max_depth = e.step_counter == v->cfg.num_layers_termination - 1;
    double rwd;                                                                // reward_fn (:484-499)
    if (e.error < e.done_threshold) rwd = 5.0;
    else if (max_depth) rwd = -5.0;
    else rwd = std::min(1.0, std::max(-1.0, (e.prev_energy - energy) / std::fabs(e.prev_energy - v->cfg.min_eig)));
    e.prev_energy = energy;
    e.rwd = rwd;
    const int energy_done = e.error < e.done_threshold;
    int d = energy_done || max_depth;
    e.previous = e.current;
    e.nfev = v->nfev[b];
    if (e.halting_step >= 0 && e.step_counter == e.halting_step) d = 1;       // rand_halt (:318-320)
    if (d) {                                                                   // (:321-324)
      e.episodes_completed += 1;
      bool ok;
      const double t = current_threshold(v, e.episodes_completed, &ok);
      if (!ok) curriculum_exhausted = true;      // reported after the loop: every environment finishes its step first
      else e.done_threshold = t;
      e.episodes_completed_saved = e.episodes_completed;
      e.lowest_energy_saved = e.lowest_energy;
    }
    reward[b] = (float)rwd;
    done[b] = d;
    if (obs_index) obs_index[b] = e.obs_index;
  }
  if (curriculum_exhausted) return fail(v, VQE_ESTATE, "curriculum: no threshold left for this episode count");
  return VQE_OK;
}

int vqe_vecenv_get(vqe_vecenv_t* v, int field, double* out) {
  if (!v || !out) return VQE_EINVAL;
  for (int b = 0; b < v->cfg.num_envs; ++b) {
    const Env& e = v->env[b];
    double val;
    switch (field) {
      case VQE_ENV_ENERGY: val = e.energy; break;
      case VQE_ENV_ERROR: val = e.error; break;
      case VQE_ENV_PREV_ENERGY: val = e.prev_energy; break;
      case VQE_ENV_NFEV: val = e.nfev; break;
      case VQE_ENV_DONE_THRESHOLD: val = e.done_threshold; break;
      case VQE_ENV_STEP_COUNTER: val = e.step_counter; break;
      case VQE_ENV_REWARD: val = e.rwd; break;
      case VQE_ENV_N_GATES: val = (double)e.gates.size(); break;
      case VQE_ENV_N_ROTATIONS: { int c = 0; for (auto& g : e.gates) c += g.kind != 0; val = c; break; }
      case VQE_ENV_LOWEST_ENERGY: val = e.lowest_energy; break;
      case VQE_ENV_EPISODES_COMPLETED: val = (double)e.episodes_completed; break;
      case VQE_ENV_HALTING_STEP: val = (double)e.halting_step; break;
      default: return fail(v, VQE_EINVAL, "unknown field");
    }
    out[b] = val;
  }
  return VQE_OK;
}

int vqe_vecenv_state(vqe_vecenv_t* v, int32_t b, float* dense) {
  if (!v || !dense || b < 0 || b >= v->cfg.num_envs) return VQE_EINVAL;
  const int n = v->cfg.n_qubits, L = v->cfg.num_layers;
  std::memset(dense, 0, sizeof(float) * (size_t)L * (n + 6) * n);
  for (const Gate& g : v->env[b].gates) {
    float* lay = dense + (size_t)g.layer * (n + 6) * n;
    if (g.kind == 0) lay[(size_t)g.q1 * n + g.q0] = 1.0f;
    else { lay[(size_t)(n + g.kind - 1) * n + g.q0] = 1.0f; lay[(size_t)(n + 3 + g.kind - 1) * n + g.q0] = g.angle; }
  }
  return VQE_OK;
}

int vqe_vecenv_moments(vqe_vecenv_t* v, int32_t b, int32_t* moments, int32_t* slots) {
  if (!v || b < 0 || b >= v->cfg.num_envs) return VQE_EINVAL;
  const int n = v->cfg.n_qubits;
  const Env& e = v->env[b];
  if (moments) for (int i = 0; i < n; ++i) moments[i] = e.moments[i];
  if (slots) for (int i = 0; i < n; ++i) for (int k = 0; k < 4; ++k) slots[4 * i + k] = e.slot_used[i] ? e.slot[i].v[k] : -1;
  return VQE_OK;
}

int vqe_vecenv_actions(vqe_vecenv_t* v, int32_t b, int32_t* current, int32_t* previous) {
  if (!v || b < 0 || b >= v->cfg.num_envs) return VQE_EINVAL;
  const Env& e = v->env[b];
  if (current) std::memcpy(current, e.current.v, sizeof e.current.v);
  if (previous) std::memcpy(previous, e.previous.v, sizeof e.previous.v);
  return VQE_OK;
}

int vqe_vecenv_opt_ang(vqe_vecenv_t* v, int32_t b, double* out, int32_t* n) {
  if (!v || !n || b < 0 || b >= v->cfg.num_envs) return VQE_EINVAL;
  const auto& o = v->env[b].opt_ang;
  *n = (int32_t)o.size();
  if (out) std::copy(o.begin(), o.end(), out);
  return VQE_OK;
}

int vqe_vecenv_last_kernel_ms(vqe_vecenv_t* v, float* ms) {
  if (!v || !ms) return VQE_EINVAL;
  return vqe_last_kernel_ms(v->eng, ms);
}

}  // extern "C"
