"""Drop-in command line: the flags, defaults, validation order, messages and output files of
/root/reference/GCI.py:897-1113, driving the HIP path in gci_amd/pipeline.py.

`-p/--plot`: the numbers of the figures come from the GPU (gci_amd/plot.py), the drawing
is matplotlib on the host as in the reference.
"""
from __future__ import annotations

import argparse
import os
import sys

from . import pipeline
from .plot import plot_depth
from .formats import bam as bamfmt
from .formats import fasta

VERSION = "GCI version 1.0"


def GCI(hifi=[], nano=[], directory=".", prefix="GCI", map_qual=30, mq_cutoff=50, iden_percent=0.9, ovlp_percent=0.9,
        clip_percent=0.1, flank_len=15, threshold=0, plot=False, depth_min=0.1, depth_max=4.0, window_size=50000,
        image_type="png", force=False, dist_percent=0.005, reference=None, regions=None, chrs=None, threads=1):
    chrs_list = []
    if chrs != None:  # noqa: E711
        chrs_list = chrs.strip().split(",")

    regions_bed = {}
    if regions != None:  # noqa: E711
        if os.path.exists(regions) and os.access(regions, os.R_OK):
            with open(regions, "r") as f:
                for line in f:
                    target, start, end = line.strip().split("\t")
                    regions_bed.setdefault(target, []).append((int(start), int(end)))
        else:
            sys.exit(f'ERROR!!! "{regions}" is not an available file')

    if directory.endswith("/"):
        directory = "/".join(directory.split("/")[:-1])
    if os.path.exists(directory):
        if not os.access(directory, os.R_OK):
            sys.exit(f'ERROR!!! The path "{directory}" is unable to read')
        if not os.access(directory, os.W_OK):
            sys.exit(f'ERROR!!! The path "{directory}" is unable to write')
    else:
        os.makedirs(directory)

    if prefix.endswith("/"):
        sys.exit(f'ERROR!!! The prefix "{prefix}" is not allowed')

    if plot == True:  # noqa: E712
        if os.path.exists(f"{directory}/images"):
            if not os.access(f"{directory}/images", os.R_OK):
                sys.exit(f'ERROR!!! The path "{directory}/images" is unable to read')
            if not os.access(f"{directory}/images", os.W_OK):
                sys.exit(f'ERROR!!! The path "{directory}/images" is unable to write')
        else:
            os.makedirs(f"{directory}/images")
        image_type = image_type.lower()

    ref_refs = fasta.record_ids_indexed(reference)
    if len(chrs_list) > 0:
        for i in chrs_list:
            if i not in ref_refs:
                sys.exit(f'ERROR!!! Chromosome "{i}" provided by `--chrs` is not in the reference')
    if len(regions_bed) > 0:
        for i in regions_bed.keys():
            if i not in ref_refs:
                sys.exit(f'ERROR!!! Chromosome "{i}" provided by `--regions` is not in the reference')
    if len(chrs_list) > 0 and len(regions_bed) > 0:
        if not all(i in chrs_list for i in regions_bed.keys()):
            sys.exit('ERROR!!! Chromosomes in the regions bed file are inconsistent with the provided list of '
                     'chromosomes\nPlease read the help message use "-h" or "--help"')

    def split(files):
        bams, pafs, refs_lengths = [], [], {}
        for file in files:
            if file.endswith(".bam"):
                bams.append(file)
                h = bamfmt.read_header(file)
                refs_lengths = {r: l for r, l in zip(h.references, h.lengths)}
            else:
                pafs.append(file)
        return bams, pafs, refs_lengths

    hifi_bam, hifi_paf, nano_bam, nano_paf = [], [], [], []
    hifi_refs_lengths, nano_refs_lengths = {}, {}
    if hifi != None:  # noqa: E711
        hifi_bam, hifi_paf, hifi_refs_lengths = split(hifi)
        if set(hifi_refs_lengths.keys()) != set(ref_refs):
            sys.exit('ERROR!!! The targets in hifi alignment files are inconsistent with the reference file\n'
                     'Please check both hifi alignment files and the reference')
    if nano != None:  # noqa: E711
        nano_bam, nano_paf, nano_refs_lengths = split(nano)
        if set(nano_refs_lengths.keys()) != set(ref_refs):
            sys.exit('ERROR!!! The targets in ont alignment files are inconsistent with the reference file\n'
                     'Please check both ont alignment files and the reference')

    print("Finding gaps ...")
    Ns_bed, Ns_bed_file = pipeline.get_Ns_ref(reference, prefix, directory, force)
    if Ns_bed_file != None:  # noqa: E711
        print(f"Finding gaps done!!! The gaps are in {Ns_bed_file}\n\n")
    else:
        print("Finding gaps done!!! Awesome! No gaps were found!\n\n")

    common = (map_qual, mq_cutoff, iden_percent, clip_percent, ovlp_percent, flank_len, directory, force)
    hint = (-1, threshold, flank_len)          # the merge_depth() scan that follows every filter()
    if nano == None:  # noqa: E711
        depths, targets_length = pipeline.filter(hifi_paf, hifi_bam, prefix, *common, "HiFi", chrs_list, threads,
                                                 issue_hint=hint)
        depths = pipeline.merge_gaps_depths(depths, Ns_bed)
        bed = pipeline.merge_depth(depths, prefix, threshold, flank_len, directory, force, "HiFi")
        pipeline.compute_index(targets_length, prefix, directory, force, [bed], ["HiFi"], flank_len, dist_percent,
                               regions_bed, [depths], threshold, chrs_list)
        if plot == True:  # noqa: E712
            plot_depth([depths], depth_min, depth_max, window_size, image_type, directory, prefix, force, targets_length,
                       dist_percent, regions_bed, threshold)
    elif hifi == None:  # noqa: E711
        depths, targets_length = pipeline.filter(nano_paf, nano_bam, prefix, *common, "ONT", chrs_list, threads,
                                                 issue_hint=hint)
        depths = pipeline.merge_gaps_depths(depths, Ns_bed)
        bed = pipeline.merge_depth(depths, prefix, threshold, flank_len, directory, force, "ONT")
        pipeline.compute_index(targets_length, prefix, directory, force, [bed], ["Nano"], flank_len, dist_percent,
                               regions_bed, [depths], threshold, chrs_list)
        if plot == True:  # noqa: E712
            plot_depth([depths], depth_min, depth_max, window_size, image_type, directory, prefix, force, targets_length,
                       dist_percent, regions_bed, threshold)
    else:
        if set(hifi_refs_lengths.keys()) != set(nano_refs_lengths.keys()):
            sys.exit('ERROR!!! The targets in hifi and nano alignment files are inconsistent\n'
                     'Please check the reference used in mapping both hifi and ont reads')
        for target, length in hifi_refs_lengths.items():
            if length != nano_refs_lengths[target]:
                sys.exit(f'ERROR!!! The element "{target}:{length}" in hifi alignment files are inconsistent with '
                         f'that in ont alignment files which is "{target}:{nano_refs_lengths[target]}"\n'
                         'Please check the reference used in mapping both hifi and ont reads')
        hifi_depths, targets_length = pipeline.filter(hifi_paf, hifi_bam, prefix + "_hifi", *common, "HiFi", chrs_list,
                                                      threads, issue_hint=hint)
        hifi_depths = pipeline.merge_gaps_depths(hifi_depths, Ns_bed)
        nano_depths, targets_length = pipeline.filter(nano_paf, nano_bam, prefix + "_nano", *common, "ONT", chrs_list,
                                                      threads, issue_hint=hint)
        nano_depths = pipeline.merge_gaps_depths(nano_depths, Ns_bed)
        two = pipeline.merge_two_type_depth(hifi_depths, nano_depths, prefix + "_two_type", directory, force, threads)
        two = pipeline.merge_gaps_depths(two, Ns_bed)
        hb = pipeline.merge_depth(hifi_depths, prefix + "_hifi", threshold, flank_len, directory, force, "HiFi")
        nb = pipeline.merge_depth(nano_depths, prefix + "_nano", threshold, flank_len, directory, force, "ONT")
        tb = pipeline.merge_depth(two, prefix + "_two_type", threshold, flank_len, directory, force, "two_types")
        pipeline.compute_index(targets_length, prefix, directory, force, [hb, nb, tb], ["HiFi", "Nano", "HiFi + Nano"],
                               flank_len, dist_percent, regions_bed, [hifi_depths, nano_depths, two], threshold,
                               chrs_list)
        if plot == True:  # noqa: E712
            plot_depth([hifi_depths, nano_depths], depth_min, depth_max, window_size, image_type, directory, prefix, force,
                       targets_length, dist_percent, regions_bed, threshold)
    print("GCI finished!!!\nBye!!!")


def build_parser(prog: str) -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser(prog=prog, add_help=False, formatter_class=argparse.RawTextHelpFormatter,
                                     description="A program for assessing the T2T genome",
                                     epilog="Examples:\npython GCI.py -r ref.fa --hifi hifi.bam hifi.paf ... "
                                            "--nano nano.bam nano.paf ...")
    io = parser.add_argument_group("Input/Output")
    io.add_argument("-r", "--reference", metavar="FILE", help="The reference file")
    io.add_argument("--hifi", nargs="+", metavar="", help="PacBio HiFi reads alignment files (at least one bam file)")
    io.add_argument("--nano", nargs="+", metavar="",
                    help="Oxford Nanopore long reads alignment files (at least one bam file)")
    io.add_argument("--chrs", metavar="", help="A list of chromosomes separated by comma")
    io.add_argument("-R", "--regions", metavar="FILE",
                    help="Bed file containing regions\nBe cautious! If both specify `--chrs` and `--regions`, "
                         "chromosomes in regions bed file should be included in the chromosomes list")
    io.add_argument("-ts", "--threshold", metavar="INT", type=int,
                    help="The threshold of depth to be reported as issues [0]", default=0)
    io.add_argument("-dp", "--dist-percent", metavar="FLOAT", type=float,
                    help="The distance between the candidate gap intervals for combining in chromosome units [0.005]",
                    default=0.005)
    io.add_argument("-t", "--threads", metavar="INT", type=int, help="Number of threads [1]", default=1)
    io.add_argument("-d", dest="directory", metavar="PATH", help="The directory of output files [.]", default=".")
    io.add_argument("-o", "--output", dest="prefix", metavar="STR", help="Prefix of output files [GCI]", default="GCI")
    fo = parser.add_argument_group("Filter Options")
    fo.add_argument("-mq", "--map-qual", metavar="INT", type=int, help="Minium mapping quality for alignments [30]",
                    default=30)
    fo.add_argument("--mq-cutoff", metavar="INT", type=int,
                    help="The cutoff of mapping quality for keeping the alignment [50]\n"
                         "(only used when inputting more than one alignment files)", default=50)
    fo.add_argument("-ip", "--iden-percent", metavar="FLOAT", type=float,
                    help="Minimum identity (num_match_res/len_aln) of alignments [0.9]", default=0.9)
    fo.add_argument("-op", "--ovlp-percent", metavar="FLOAT", type=float,
                    help="Minimum overlapping percentage of the same read alignment if inputting more than one "
                         "alignment files [0.9]", default=0.9)
    fo.add_argument("-cp", "--clip-percent", metavar="FLOAT", type=float,
                    help="Maximum clipped percentage of the alignment [0.1]", default=0.1)
    fo.add_argument("-fl", "--flank-len", metavar="INT", type=int,
                    help="The flanking length of the clipped bases [15]", default=15)
    po = parser.add_argument_group("Plot Options")
    po.add_argument("-p", "--plot", action="store_const", const=True, default=False,
                    help="Visualize the finally filtered whole genome (and regions if providing the option `-R`) "
                         "depth [False]")
    po.add_argument("-dmin", "--depth-min", metavar="FLOAT", type=float,
                    help="Minimum depth in folds of mean coverage for plotting [0.1]", default=0.1)
    po.add_argument("-dmax", "--depth-max", metavar="FLOAT", type=float,
                    help="Maximum depth in folds of mean coverage for plotting [4.0]", default=4.0)
    po.add_argument("-ws", "--window-size", metavar="INT", type=int, help="The window size when plotting [50000]",
                    default=50000)
    po.add_argument("-it", "--image-type", metavar="STR", help="The format of the output images: png or pdf [png]",
                    default="png")
    op = parser.add_argument_group("Other Options")
    op.add_argument("-f", "--force", action="store_const", const=True, default=False,
                    help="Force rewriting of existing files [False]")
    op.add_argument("-h", "--help", action="help", help="Show this help message and exit")
    op.add_argument("-v", "--version", action="version", version=VERSION,
                    help="Show program's version number and exit")
    return parser


def _check_inputs(files, what):
    bam_num = 0
    for file in files:
        if os.path.exists(file) and os.access(file, os.R_OK):
            if file.endswith(".bam"):
                bam_num += 1
        else:
            sys.exit(f'ERROR!!! "{file}" is not an available file')
    if bam_num == 0:
        sys.exit(f'ERROR!!! Please input at least one {what} bam file\n'
                 'Please read the help message use "-h" or "--help"')


def main(argv=None):
    argv = sys.argv if argv is None else argv
    parser = build_parser(argv[0])
    args = vars(parser.parse_args(argv[1:]))
    if len(argv) == 1:
        parser.print_help()
        sys.exit()
    if (args["hifi"] == None) and (args["nano"] == None):  # noqa: E711
        sys.exit('ERROR!!! Please input at least one type of TGS reads alignment files (PacBio HiFi and/or Oxford '
                 'Nanopore long reads)\nPlease read the help message use "-h" or "--help"')
    if args["hifi"] != None:  # noqa: E711
        _check_inputs(args["hifi"], "PacBio HiFi reads")
    if args["nano"] != None:  # noqa: E711
        _check_inputs(args["nano"], "Oxford Nanopore long reads")
    if args["reference"] == None:  # noqa: E711
        sys.exit('ERROR!!! Please input the reference file\nPlease read the help message use "-h" or "--help"')
    elif not (os.path.exists(args["reference"]) and os.access(args["reference"], os.R_OK)):
        sys.exit(f'ERROR!!! "{args["reference"]}" is not an available file')
    if args["map_qual"] > args["mq_cutoff"]:
        print(f'WARNING!!! The minium mapping quality ({args["map_qual"]}) is higher than the cutoff '
              f'({args["mq_cutoff"]}), which means that wouldn\'t filter any reads\n'
              'Please read the help message use "-h" or "--help"', file=sys.stderr)
    print(f"Used arguments:{args}")
    GCI(**args)


if __name__ == "__main__":
    main()
