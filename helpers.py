"""The reference's helpers module surface (reference helpers.py), served by ai-font-renderer_amd/helpers.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ai_font_renderer_amd.helpers import *  # noqa: E402,F401,F403
from ai_font_renderer_amd.helpers import MODEL_FILENAME  # noqa: E402,F401
