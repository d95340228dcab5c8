"""Names of the decoder's phase slots in the diagnostic build (scripts/gpu_phase_dec*.py)."""
names = ['0 before fill barrier', '1 wait: fill barrier', '2 copier: out-of-buffer matches', '3 copier: matches', '4 copier: flush',
         '5 parser: chain + lane fields', '6 parser: literals', '7 wait: batch barrier', '8 copier: slide', '9 wait: slide barrier',
         '10 copier: general loop', '11 wait: general barrier', '12', '13', '14', '15 total cycles',
         '16 parse: window', '17 parse: four-chunk steps', '18 parse: single-token steps', '19 parse: lane fields', '20 parse: places and checks',
         '21 count: four-chunk steps', '22 count: single-token steps', '23 count: parses', '24 parse: one-chunk steps', '25 count: one-chunk steps']
