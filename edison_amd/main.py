#!/usr/bin/env python
"""``./main.py <module> <command>`` -- the reference's CLI (audio/main.py:12-39) for the modules that sit on the
keyword-spotting hot path: ``mfcc host`` / ``mfcc mcu <mode>`` (main.py:80-99,199-205) and ``kws mcu <mode> [file]`` /
``kws live ...`` (main.py:146-165,231-238). Same dispatch pattern. Where the reference talks to the STM32 board over
the UART, the board's leg is computed by the GPU's bit-exact variant C. The other modules (mic, acquire, train,
deploy, mcu) drive the board or Keras training and are out of scope here.
"""
import argparse
import sys


class Edison(object):

  def __init__(self, argv=None):
    self.argv = list(sys.argv if argv is None else argv)
    self.rc = 0
    parser = argparse.ArgumentParser(
      description='Edison Keyword Spotting tools (MI355X hot path)',
      usage='''edison <module> <command> [<args>]

The modules:
    mfcc      Mel frequency cepstral coefficient tools
    kws       Experiment with the keyword spotting algorithm
''')
    parser.add_argument('module', help='Which module to select')
    args = parser.parse_args(self.argv[1:2])
    if args.module.startswith('_') or not hasattr(self, args.module) or args.module in ('argv', 'rc'):
      print('Unrecognized module')
      parser.print_help()
      self.rc = 1
      return
    getattr(self, args.module)()

  def mfcc(self):
    parser = argparse.ArgumentParser(usage='''edison mfcc <command> [<args>]

Commands
    host    Run MFCC on host (MI355X)
    mcu     The board comparisons with the GPU's Q15 MFCC as the board: mcu calc [file] | single [wav] | file <wav>
''')
    parser.add_argument('command', help='Command to run')
    args = parser.parse_args(self.argv[2:3])
    if args.command == 'host':
      self.mfcc_host()
    elif args.command == 'mcu':
      self.mfcc_mcu()
    else:
      print('Unrecognized command')
      parser.print_help()
      self.rc = 1

  def kws(self):
    parser = argparse.ArgumentParser(usage='''edison kws <command> [<args>]

Commands
    mcu     KWS on a wav file: mcu file <wav> | mcu frame <wav>
    live    The firmware's continuous mode replayed on a wav: live host <wav> | live mcu <wav>
''')
    parser.add_argument('command', help='Command to run')
    args = parser.parse_args(self.argv[2:3])
    if args.command == 'live':
      self.kws_live()
    elif args.command == 'mcu':   # the reference falls through to "Unrecognized command" after `live` (main.py:158-165)
      self.kws_mcu()
    else:
      print('Unrecognized command')
      parser.print_help()
      self.rc = 1

  ######################################################
  # Final commands to run

  def mfcc_host(self):
    from edison_amd.mfcc import mfcc as mfcc_script
    self.rc = mfcc_script.main(self.argv[3:])

  def mfcc_mcu(self):
    from edison_amd.mfcc import mfcc_on_mcu
    self.rc = mfcc_on_mcu.main(self.argv[2:])

  def kws_live(self):
    from edison_amd.kws import kws_live
    self.rc = kws_live.main(self.argv[2:])

  def kws_mcu(self):
    from edison_amd.kws import kws_host
    self.rc = kws_host.main(self.argv[2:])


def main(argv=None):
  return Edison(argv).rc


if __name__ == '__main__':
  sys.exit(main())
