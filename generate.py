#!/usr/bin/env python
"""`python generate.py <checkpoint_dir> --mel m.npy --gc_cardinality 2 --gc_id 0` -- the reference's CLI on MI355X."""
import twvk_amd  # noqa: F401  (registers the hyphenated package directory)
from twvk_amd.generate import main

if __name__ == "__main__":
    import time
    s = time.time()
    main()
    print(time.time() - s, 'sec')
