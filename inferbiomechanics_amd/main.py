"""InferBiomechanics command line (drop-in for src/main.py:16-58): `python3 main.py {train,analyze,visualize} ...`.

Commands on the GPU hot path are implemented; the reference's data-preparation / plotting / GUI commands
(create-splits, pickle-data, sanity-check, make-plots, review-file, visualize-file, save-prediction-csv) are
nimblephysics / matplotlib housekeeping outside that path and are reported as such."""
import argparse
import logging
import os
import sys

if __package__ in (None, ""):      # allow `python3 main.py ...` from inside the package directory
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    __package__ = "inferbiomechanics_amd"

from .cli.analyze import AnalyzeCommand      # noqa: E402
from .cli.train import TrainCommand          # noqa: E402
from .cli.visualize import VisualizeCommand  # noqa: E402

OUT_OF_SCOPE = ['visualize-file', 'create-splits', 'sanity-check', 'make-plots', 'review-file', 'pickle-data',
                'save-prediction-csv']


def main(argv=None):
    commands = [TrainCommand(), VisualizeCommand(), AnalyzeCommand()]
    parser = argparse.ArgumentParser(description='InferBiomechanics Command Line Interface (MI355X hot path)')
    subparsers = parser.add_subparsers(dest="command")
    for command in commands:
        command.register_subcommand(subparsers)
    for name in OUT_OF_SCOPE:
        subparsers.add_parser(name, help='(reference housekeeping command; outside the GPU hot path)')
    args = parser.parse_args(argv)
    if args.command in OUT_OF_SCOPE:
        raise SystemExit(f"`{args.command}` is a nimblephysics/matplotlib housekeeping command of the reference and "
                         f"is outside the MI355X hot path; use the reference implementation for it.")
    for command in commands:
        if command.run(args):
            return True
    return False


def _setup_logging():
    logging.basicConfig(filename="log", format='%(asctime)s %(message)s', filemode='a')
    logger = logging.getLogger()
    logger.addHandler(logging.StreamHandler())
    logger.setLevel(logging.INFO)


if __name__ == '__main__':
    _setup_logging()
    main()
