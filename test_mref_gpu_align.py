#!/usr/bin/env python3
"""Drop-in for the reference's entry point of the same name (multi-reference 2-D alignment);
see cryo_ralib_amd/cli.py.  Run one process per GPU with torch.distributed.run."""
import sys
from cryo_ralib_amd.cli import main_mref

if __name__ == "__main__":
    sys.exit(main_mref())
