"""mains/main_ceVAE.py of the reference: the `ceVAE` trainer on `models/context_encoder_variational_autoencoder.py` -- here the same pairing through run.py's driver
(all of run.py's flags apply; `python mains/main_ceVAE.py -E 10 -b 64`)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from run import build_parser, main  # noqa: E402

if __name__ == '__main__':
    ap = build_parser()
    ap.set_defaults(trainer='ceVAE', model='context_encoder_variational_autoencoder')
    args = ap.parse_args()
    main(args)
