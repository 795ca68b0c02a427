// tests/cpp/timing_csv_check.cpp — re-emits rows through include/hnet_timing_csv.h (tests/test_timing_csv.py):
//   timing_csv_check <out.csv> < "t load prop nn upd total" lines on stdin
#include <cstdio>
#include "hnet_timing_csv.h"

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    hnet_csv::TimingCsv csv;
    if (!csv.open(argv[1])) return 3;
    double v[6];
    while (std::scanf("%lf %lf %lf %lf %lf %lf", &v[0], &v[1], &v[2], &v[3], &v[4], &v[5]) == 6) csv.append(v[0], v[1], v[2], v[3], v[4], v[5]);
    return 0;
}
