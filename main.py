#!/usr/bin/env python3
"""
 multi-gpu, multi-server distributed learning on MI355X: same command line as the reference
 (/root/reference/main.py:20-58):

   python main.py train -d DATA_PATH [-b N] [-e N] [-f CHECKPOINT] [--debug]
   python main.py test  -d DATA_PATH -f CHECKPOINT [-b N] [--debug]

 One process is spawned per GPU of this node (config.DDTNodes); run the same command on every node listed there.
"""
import sys

from distributedpytorch_amd.launcher import main

if __name__ == "__main__":
    sys.exit(main())
