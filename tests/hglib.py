"""Loads the product package (hyper-greco_amd/, C ABI over libhypergreco.so) for the tests."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

hg = entry.load_package()


def have_gpu():
    try:
        return hg.lib().hg_device_count() > 0
    except Exception:
        return False
