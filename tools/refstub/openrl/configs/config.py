import argparse


def create_config_parser():
    return argparse.ArgumentParser()
