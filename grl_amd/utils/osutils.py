import os


def mkdir_if_missing(dir_path):
    os.makedirs(dir_path, exist_ok=True)
