from .groups import SE3, SO3, LieGroup, RxSO3, Sim3, cat, stack  # noqa: F401
