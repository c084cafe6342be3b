"""The constraint masks as SURVEY.md section 8a extracted them from the reference's source (a regex pass over the
`constraints` bodies of layouts/src/{recursive,starknet}/air.rs): the checksum the restated AIRs are held to."""

# recursive: the full mask (column: row offsets), 133 cells
RECURSIVE_MASK = {
    0: list(range(16)),
    1: [0, 1] + list(range(2, 33, 2)) + [33, 64, 65, 88, 90, 92, 94, 96, 97, 120, 122, 124, 126],
    2: [0, 1],
    3: [0, 1, 2, 3, 4, 5, 8, 9, 10, 11, 12, 13, 16, 26, 27, 42, 43, 58, 74, 75, 91, 122, 123, 154, 202, 522, 523,
        1034, 1035, 2058],
    4: [0, 1, 2, 3],
    5: list(range(9)) + [12, 28, 44, 60, 76, 92, 108, 124, 1021, 1023, 1025, 1027, 2045],
    6: [0, 1, 2, 3, 4, 5, 7, 9, 11, 13, 17, 25, 768, 772, 784, 788, 1004, 1008, 1022, 1024],
    7: [0, 1], 8: [0, 1], 9: [0, 1, 2, 5],
}
# starknet: cells per column (269 in total) and the largest row offset per column
STARKNET_CELLS_PER_COLUMN = [16, 5, 4, 9, 2, 60, 4, 56, 105, 8]
STARKNET_MAX_OFFSET = [15, 511, 256, 256, 255, 33158, 3, 1009, 32763, 15]
