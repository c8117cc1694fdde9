"""mains/main_GMVAE.py of the reference: the `GMVAE` trainer on `models/gaussian_mixture_variational_autoencoder.py` -- here the same pairing through run.py's driver
(all of run.py's flags apply; `python mains/main_GMVAE.py -E 10 -b 64`)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from run import build_parser, main  # noqa: E402

if __name__ == '__main__':
    ap = build_parser()
    ap.set_defaults(trainer='GMVAE', model='gaussian_mixture_variational_autoencoder')
    args = ap.parse_args()
    main(args)
