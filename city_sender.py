#!/usr/bin/env python3
"""Drop-in entry point with the reference CLI's name and flags (reference city_sender.py:47-223, 467-617);
the implementation is evc_amd/cli.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import evc_amd  # noqa: E402,F401
from evc_amd.cli import main  # noqa: E402

if __name__ == "__main__":
    main()
