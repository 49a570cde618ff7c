from .VSLNet import VSLNet, build_optimizer_and_scheduler  # noqa: F401
