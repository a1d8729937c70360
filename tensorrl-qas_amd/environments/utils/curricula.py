"""Done-threshold curricula with the reference's semantics
(environments/utils/curricula.py:2-98), restated; pinned by tests/golden/host_logic.json."""


class VanillaCurriculum:
    """Piecewise-constant threshold: thresholds[i] while fewer than switch_episodes[i]
    episodes have finished."""

    def __init__(self, config, **kw):
        self.thresholds = config["thresholds"]
        self.episodes = config["switch_episodes"]
        self.episodes_completed = 0
        self.min_en = kw.get("target_energy")
        self.current_threshold = config["accept_err"]
        self.lowest_energy = self.min_en + self.current_threshold

    def get_current_threshold(self):
        pending = [i for i, e in enumerate(self.episodes) if e > self.episodes_completed]
        return self.thresholds[min(pending)]

    def update_threshold(self, **kw):
        self.episodes_completed += 1


class SuccesCountThreshold:
    def __init__(self, config, **kw):
        self.min_en = kw.get("target_energy")
        self.success_thresh = config["success_thresh"]
        self.current_threshold = config["accept_err"]
        self.lowest_energy = self.min_en + self.current_threshold
        self.success_counter = 0

    def greedy_shift(self):
        if self.success_thresh:
            self.success_counter += 1
            if self.success_counter >= self.success_thresh:
                self.success_counter = 0
                self.current_threshold = abs(self.min_en - self.lowest_energy)
        return self.current_threshold

    def get_current_threshold(self):
        return self.current_threshold

    def update_threshold(self, **kw):
        if kw.get("energy_done"):
            self.greedy_shift()


class MovingThreshold:
    def __init__(self, config, **kw):
        self.amortisation = config["shift_threshold_ball"]
        self.greedy_shift_time = config["shift_threshold_time"]
        self.min_en = kw.get("target_energy")
        self.success_thresh = config["success_thresh"]
        self.succ_radius_shift = config["succ_radius_shift"]
        self.succes_switch = config["succes_switch"]
        self.current_threshold = config["accept_err"]
        self.lowest_energy = self.min_en + self.current_threshold
        self.success_counter = 0
        self.radius_shift_counter = 0
        self.call_counter = 0

    def reduce_amortisation(self):
        if self.success_thresh:
            self.success_counter += 1
            gap = abs(self.min_en - self.lowest_energy)
            if (self.success_counter >= self.success_thresh and self.radius_shift_counter < self.succ_radius_shift
                    and self.succes_switch > gap):
                self.current_threshold -= self.amortisation / self.succ_radius_shift
                self.success_counter = 0
                self.radius_shift_counter += 1
        return self.current_threshold

    def greedy_shift(self):
        self.call_counter += 1
        if self.call_counter > 10 and self.call_counter % self.greedy_shift_time == 0:
            gap = abs(self.min_en - self.lowest_energy)
            if self.amortisation:
                self.current_threshold = gap + self.amortisation
                if self.success_thresh:
                    self.radius_shift_counter = 0
                    self.success_counter = 0
            else:
                self.current_threshold = gap
        return self.current_threshold

    def get_current_threshold(self):
        return self.current_threshold

    def update_threshold(self, **kw):
        if kw.get("energy_done"):
            self.reduce_amortisation()
        self.greedy_shift()
