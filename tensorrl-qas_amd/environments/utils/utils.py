"""Config parsing and action tables with the reference's semantics
(environments/utils/utils.py:6-78), restated; pinned by tests/golden/host_logic.json."""
import configparser
import json
from itertools import product

_FLOAT_KEYS = frozenset(("learning_rate", "dropout", "alpha", "beta", "beta_incr", "shift_threshold_ball",
                         "succes_switch", "tolearance_to_thresh", "memory_reset_threshold",
                         "fake_min_energy", "_true_en"))
_STR_KEYS = frozenset(("ham_type", "fn_type", "geometry", "method", "agent_type", "agent_class", "init_seed",
                       "init_path", "init_thresh", "mapping", "optim_alg", "curriculum_type"))
_JSON_KEYS = frozenset(("episodes", "neurons", "accept_err", "epsilon_decay", "epsilon_min", "final_gamma",
                        "memory_clean", "update_target_net", "epsilon_restart", "thresholds", "switch_episodes"))


def _coerce(key, raw):
    """int() first, else the raw string; then the key decides (float / str / json)."""
    if key in _FLOAT_KEYS:
        return float(raw)
    if key in _STR_KEYS:
        return str(raw)
    if key in _JSON_KEYS:
        return json.loads(raw)
    try:
        return int(raw)
    except ValueError:
        return raw


def get_config(config_name, experiment_name, path="configuration_files", verbose=True):
    """{section: {key: value}} of ``{path}/{config_name}{experiment_name}`` (the reference's
    odd argument order is kept: it is called as get_config(name, '.cfg', path=...))."""
    parser = configparser.ConfigParser()
    parser.read("{}/{}{}".format(path, config_name, experiment_name))
    return {sec: {k: _coerce(k, v) for k, v in parser.items(sec)} for sec in parser.sections()}


def dictionary_of_actions(num_qubits):
    """index -> [ctrl, offset, rot_qubit, rot_axis]: n(n-1) CNOT actions (target =
    (ctrl+offset) mod n) then 3n rotations (axis 1,2,3 = X,Y,Z); the value ``num_qubits``
    marks "no such gate"."""
    n = num_qubits
    table = [[c, x, n, 0] for c, x in product(range(n), range(1, n))]
    table += [[n, 0, r, h] for r, h in product(range(n), range(1, 4))]
    return dict(enumerate(table))


def dict_of_actions_revert_q(num_qubits):
    n = num_qubits
    table = [[c, x, n, 0] for c, x in product(range(n - 1, -1, -1), range(n - 1, 0, -1))]
    table += [[n, 0, r, h] for r, h in product(range(n - 1, -1, -1), range(1, 4))]
    return dict(enumerate(table))
