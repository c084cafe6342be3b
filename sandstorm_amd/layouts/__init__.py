"""Restatements of the reference's layouts (layouts/src/*): AIR constraints as air_program expressions and the
host-side base-trace generation they are checked against ("next" row X1)."""
