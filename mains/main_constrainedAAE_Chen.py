"""mains/main_constrainedAAE_Chen.py of the reference: the `ConstrainedAAE` trainer on `models/constrained_adversarial_autoencoder_Chen.py` -- here the same pairing through run.py's driver
(all of run.py's flags apply; `python mains/main_constrainedAAE_Chen.py -E 10 -b 64`)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from run import build_parser, main  # noqa: E402

if __name__ == '__main__':
    ap = build_parser()
    ap.set_defaults(trainer='ConstrainedAAE', model='constrained_adversarial_autoencoder_Chen')
    args = ap.parse_args()
    if args.intermediateResolutions == (8, 8) and args.outputHeight // 8 != 8:      # this graph's latent map is height / 8
        args.intermediateResolutions = (args.outputHeight // 8, args.outputWidth // 8)
    main(args)
