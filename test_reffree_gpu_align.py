#!/usr/bin/env python3
"""Drop-in for the reference's entry point of the same name (reference-free 2-D alignment);
see cryo_ralib_amd/cli.py."""
import sys
from cryo_ralib_amd.cli import main_reffree

if __name__ == "__main__":
    sys.exit(main_reffree())
