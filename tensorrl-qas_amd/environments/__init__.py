"""Environment layer: same module and class names as the reference's ``environments`` package
(environment_qulacs*.py -> CircuitEnv) so a driver only changes its import line."""
