#!/usr/bin/env python3
"""`python model.py --train` / `python model.py`: the reference's command line (reference model.py:425-454),
served by the MI355X-native implementation in ai-font-renderer_amd/."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ai_font_renderer_amd.model import *  # noqa: E402,F401,F403
from ai_font_renderer_amd import model as _impl  # noqa: E402

if __name__ == "__main__":
    _impl.main(sys.argv)
