"""Re-run the two tolerance-edge shapes of the round-4 fuzz sweep (seed 2026: fuzz 17, fuzz 53); with VSLNET_HIP_LIB=<base build> this tells a
regression from a pre-existing cancellation artefact."""
import sys
sys.path.insert(0, '.')
from tests.test_hip_training import test_baseline_shapes_against_oracle as check  # noqa: E402
for shape in (dict(name='fuzz 17', B=1, T=4, Lq=65, Lc=17, Dv=1024, char_dim=64, char_size=40, word_table=False),
              dict(name='fuzz 53', B=3, T=4, Lq=32, Lc=10, Dv=500, char_dim=50, char_size=40, word_table=False)):
    try:
        check(shape)
        print('ok  ', shape['name'])
    except AssertionError as e:
        print('FAIL', shape['name'], repr(e)[:300])
