"""`model/` -- the reference's import path (`from model.VSLNet import VSLNet`) kept as a thin alias of
vslnet_amd.model so existing callers drop in (the reference's PyTorch files carry a `_t7` suffix; its README calls
them main.py / model/VSLNet.py -- SURVEY.md naming note)."""
