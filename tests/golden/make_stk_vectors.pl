#!/usr/bin/perl
# Golden vectors for the wrapper logic between a Stockholm seed alignment and the RAMExtend command line
# (reference util/extend-stk.pl): the "extendable" flags of every row (:330-346, with the reference's own two regular expressions,
# the start-1 of :314 and the gi|NNN rule of :308-311) on the reference's three fixtures test/ce10-fam{1,2,3}.stk, and the
# divergence -> matrix / -minimprovement ladder of :291-302 on a list of divergences around its thresholds.
# RepeatModeler's Perl modules (the wrapper's Stockholm parser) are not part of the reference, so rows are read here from the file
# format directly: "<[assembly:]sequence:start-end[_orient]> <aligned>" lines.  Run: perl tests/golden/make_stk_vectors.pl > tests/golden/stk_vectors.json
use strict; use warnings;
my $dir = "tests/golden/inputs";
my @fams = ("ce10-fam1.stk", "ce10-fam2.stk", "ce10-fam3.stk");
my @out;
foreach my $f (@fams) {
  open(my $fh, "<", "$dir/$f") or die "$dir/$f: $!";
  my (@rows, $extendable_count, %seen, @order, $desc);
  $extendable_count = 0; $desc = "";
  while (my $line = <$fh>) {
    chomp $line;
    if ($line =~ /^#=GF\s+DE\s+(.*)$/) { $desc .= ($desc eq "" ? "" : " ") . $1; next; }
    next if $line =~ /^#/ || $line =~ /^\/\// || $line =~ /^\s*$/;
    my ($name, $sequence) = split(/\s+/, $line);
    next unless defined $sequence;
    if (!exists $seen{$name}) { $seen{$name} = ""; push @order, $name; }
    $seen{$name} .= $sequence;
  }
  close $fh;
  foreach my $name (@order) {
    my $sequence = $seen{$name};
    $name =~ /^(?:(\S+):)?([^:\s]+):(\d+)-(\d+)(?:_([+-]))?$/ or die "row name $name";
    my ($sequenceName, $start, $end, $orient) = ($2, $3, $4, $5);
    if (!defined $orient) { $orient = "+"; if ($start > $end) { ($start, $end, $orient) = ($end, $start, "-"); } }
    next if $sequenceName =~ /^gi\|\d+$/;                                   # extend-stk.pl:308-311
    $start--;                                                              # :314 zero-based half open
    my ($l, $r);
    if ( $sequence =~ /^[\.]{0,10}[^\.]/ && $sequence =~ /[^\.][\.]{0,10}$/ ) { ($l, $r) = (1, 1); $extendable_count++; }   # :330-333
    elsif ( $sequence =~ /^[\.]{0,10}[^\.]/ ) { ($l, $r) = (1, 0); $extendable_count++; }                                  # :334-338
    elsif ( $sequence =~ /[^\.][\.]{0,10}$/ ) { ($l, $r) = (0, 1); $extendable_count++; }                                  # :339-343
    else { ($l, $r) = (0, 0); }                                                                                            # :344-346
    push @rows, "[\"$sequenceName\", $start, $end, $l, $r, \"$orient\"]";
  }
  my $mdiv = ($desc =~ /mDiv=(\d+\.\d+)/) ? $1 : "null";                   # :286-289
  push @out, "  {\"file\": \"$f\", \"extendable_count\": $extendable_count, \"runs_ramextend\": " . ($extendable_count > 3 ? "true" : "false") .
             ", \"mDiv\": $mdiv, \"rows\": [\n    " . join(",\n    ", @rows) . "]}";
}
my @ladder;
foreach my $tdiv (0, 5.5, 13.99, 14, 15.99, 16, 16.01, 18.99, 19, 19.5, 22.49, 22.5, 22.51, 30, 45.2) {
  foreach my $min_aligning_seqs (1, 3, 4) {
    my $div = 14; my $diagAvg = 10;                                        # :291-300
    if ($tdiv >= 16) { $div = 18; if ($tdiv >= 19) { $div = 20; if ($tdiv >= 22.5) { $div = 25; $diagAvg = 9; } } }
    my $minimprovement = $min_aligning_seqs * $diagAvg;                    # :302
    push @ladder, "[$tdiv, $min_aligning_seqs, \"${div}p43g\", $minimprovement]";
  }
}
print "{\"generator\": \"tests/golden/make_stk_vectors.pl (perl, the reference's regular expressions and thresholds verbatim)\",\n \"families\": [\n" .
      join(",\n", @out) . "],\n \"scoring_ladder\": [" . join(", ", @ladder) . "]}\n";
