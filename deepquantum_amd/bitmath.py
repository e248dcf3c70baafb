"""Integer bit helpers for the index-bit-partitioned state (mirrors the helper set of the reference's
bitmath.py:1-55; arXiv:2311.01512 Alg. 1).  Work on Python ints and on integer tensors alike."""

from __future__ import annotations


def power_of_2(exp: int) -> int:
    return 1 << exp


def is_power_of_2(number: int) -> bool:
    return number > 0 and (number & (number - 1)) == 0


def log_base2(number: int) -> int:
    assert is_power_of_2(number), f'{number} is not a power of two'
    return number.bit_length() - 1


def get_bit(number, bit_index: int):
    return (number >> bit_index) & 1


def flip_bit(number, bit_index: int):
    return number ^ (1 << bit_index)


def flip_bits(number, bit_indices):
    for b in bit_indices:
        number = number ^ (1 << b)
    return number


def insert_bit(number, bit_index: int, bit_value: int):
    low = number & ((1 << bit_index) - 1)
    return ((number >> bit_index) << (bit_index + 1)) | (bit_value << bit_index) | low


def all_bits_are_one(number: int, bit_indices) -> bool:
    return all((number >> b) & 1 for b in bit_indices)


def get_bit_mask(bit_indices) -> int:
    mask = 0
    for b in bit_indices:
        mask |= 1 << b
    return mask
