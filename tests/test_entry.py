"""GPU: the driver's smoke entry point (__graft_entry__.smoke) is part of the suite, so a change of semantics that
breaks it (round 3: one de-emphasis state per channel, shared by the batched and the per-channel caller) shows up
here and not only at the end of a round."""

import pytest

from conftest import have_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not have_gpu(), reason="needs an MI355X")]


def test_smoke_entry_point(capsys):
    import __graft_entry__ as entry
    entry.smoke()
    assert "smoke ok" in capsys.readouterr().out
