// KmerDistribution.h -- histogram of k-mer frequencies with the five-number summary `stride kmercheck` prints
// (the part of the reference's Util/KmerDistribution.{h,cpp}:25-31,86-153 that kmercheck uses: add, +=, computeKDAttributes,
// operator<<, compare).
#pragma once
#include <cmath>
#include <map>
#include <ostream>

namespace stride {

class KmerDistribution {
public:
    void add(int kmerFreq) { ++m_data[kmerFreq]; ++m_total; }
    void operator+=(const KmerDistribution& other)
    {
        for(const auto& kv : other.m_data) m_data[kv.first] += kv.second;
        m_total += other.m_total;
    }
    int getTotalKmers() const { return m_total; }

    // Quartiles by cumulative count (a quartile is the LAST frequency whose [before, after] count range contains the rank, so
    // ties move it up), whiskers at 1.5 IQR (min: first frequency >= q1 - 1.5 IQR, taken only while still 0; max: the
    // frequency just below the first one above q3 + 1.5 IQR, else the largest).  Nothing is reset between calls, as in the
    // reference (computeKDAttributes, KmerDistribution.cpp:86-132).
    void computeKDAttributes()
    {
        const int low = m_total * 1 / 4, mid = m_total * 2 / 4, upp = m_total * 3 / 4;
        int before = 0, after = 0, most = 0;
        for(const auto& kv : m_data) {
            if(kv.second > most) { most = kv.second; m_mode = kv.first; }
            before = after;
            after += kv.second;
            if(low >= before && low <= after) m_q1 = kv.first;
            if(mid >= before && mid <= after) m_q2 = kv.first;
            if(upp >= before && upp <= after) m_q3 = kv.first;
        }
        const int iqr = m_q3 - m_q1;
        const int small = m_q1 - (int)(iqr * 1.5), large = m_q3 + (int)(iqr * 1.5);
        int prev = 0, curr = 0;
        for(const auto& kv : m_data) {
            prev = curr;
            curr = kv.first;
            if(m_min == 0 && curr >= small) m_min = curr;
            if(prev <= large && curr > large) m_max = prev;
        }
        if(m_max == 0) m_max = curr;
        int sqsum = 0;
        for(const auto& kv : m_data) sqsum += kv.second * std::pow((kv.first - m_q2), 2);      // int += double, truncating each time
        m_sdv = std::sqrt((double)sqsum / (m_total - 1));
    }

    friend std::ostream& operator<<(std::ostream& out, const KmerDistribution& o)
    {
        return out << o.m_min << ' ' << o.m_q1 << ' ' << o.m_q2 << ' ' << o.m_q3 << ' ' << o.m_max;
    }

    // total.box: "cov k | <wrong k-mers> | <correct k-mers>";  value.box: "cov k <threshold>" where the threshold is the
    // correct k-mers' lower whisker if it clears the wrong ones' upper whisker, else their first quartile (.cpp:140-153)
    friend void compare(std::ostream& t, std::ostream& v, int cov, int ksize, KmerDistribution& c, KmerDistribution& e)
    {
        c.computeKDAttributes();
        e.computeKDAttributes();
        t << cov << ' ' << ksize << " | " << e << " | " << c << '\n';
        const int value = c.m_min >= e.m_max ? c.m_min : c.m_q1;
        v << cov << ' ' << ksize << ' ' << value << '\n';
    }

private:
    std::map<int, int> m_data;
    int m_total = 0, m_q1 = 0, m_q2 = 0, m_q3 = 0, m_min = 0, m_max = 0, m_mode = 0;
    double m_sdv = 0;
};

} // namespace stride
