from vslnet_amd.model.layers import *  # noqa: F401,F403
