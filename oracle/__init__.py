"""TEST INFRASTRUCTURE — CPU oracle of the PuzzleFusion++ denoise-and-verify path.

Nothing under oracle/ is imported by the product package; see oracle/pfpp_oracle.py.
"""
