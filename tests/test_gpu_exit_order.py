"""Round 2 saw `bench.py --mode infer` print its line and die with SIGSEGV at exit on an intermediate tree, and hid it with a
`del` ordering in the caller.  The library must not depend on the order in which a host drops its objects: this runs the inference
bench's flow in child processes (tools/exit_order_probe.py) that drop the handle FIRST — while the torch wrapper of its stream,
the pinned label buffer and the device tensors are still alive, with and without a device synchronisation before — and that drop
nothing explicitly; every child must leave with 0."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("variant", ["natural", "net_first_nosync", "handle_last"])
def test_the_order_in_which_a_host_drops_its_objects_does_not_matter(variant):
    r = subprocess.run([sys.executable, "-X", "faulthandler", os.path.join(ROOT, "tools", "exit_order_probe.py"), variant], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, (r.returncode, r.stderr[-1500:])
    assert "flow returned" in r.stdout
