// Known answers of mjpc/test/spline/spline_test.cc for the host TimeSpline.
#include "mjpc/spline/spline.h"

#include <cmath>

#include "check.h"
using namespace mjpc::spline;

int main() {
  for (auto interp : {kZeroSpline, kLinearSpline, kCubicSpline}) {
    {  // Empty / OneNode / TwoNodes (spline_test.cc:41-83)
      TimeSpline s(10);
      CHECK(s.Size() == 0 && s.Dim() == 10);
      for (double v : s.Sample(2.0)) CHECK(v == 0.0);
      TimeSpline a(2);
      a.SetInterpolation(interp);
      CHECK(a.Interpolation() == interp);
      a.AddNode(1.0, {1.0, 2.0});
      for (double t : {0.0, 2.0, 4.0}) CHECK(Eq(a.Sample(t), {1.0, 2.0}));
      TimeSpline::Node n = a.AddNode(2.0);
      n.values()[0] = 3.0;
      n.values()[1] = 4.0;
      CHECK(a.Size() == 2);
      CHECK(Eq(a.Sample(0), {1.0, 2.0}) && Eq(a.Sample(1), {1.0, 2.0}));
      CHECK(Eq(a.Sample(2), {3.0, 4.0}) && Eq(a.Sample(3), {3.0, 4.0}));
    }
    {  // DiscardBefore (:187-231)
      TimeSpline s(2);
      s.SetInterpolation(interp);
      for (int k = 1; k <= 4; k++) s.AddNode(k, {(double)k, (double)k + 1});
      CHECK(s.DiscardBefore(0.9) == 0 && s.Size() == 4);
      const int n = s.DiscardBefore(3.0);
      if (interp == kCubicSpline) { CHECK(n == 1 && s.Size() == 3 && Eq(s.Sample(1.0), {2.0, 3.0})); }
      else { CHECK(n == 2 && s.Size() == 2 && Eq(s.Sample(1.0), {3.0, 4.0})); }
      CHECK(s.DiscardBefore(3.9) == 0);
    }
  }
  {  // AddNodeBeforeStart (:85-100) and the middle-insert check (spline.cc:217-219)
    TimeSpline s(2);
    s.AddNode(2.0, {2.0, 3.0}); s.AddNode(1.0, {1.0, 2.0}); s.AddNode(3.0, {3.0, 4.0}); s.AddNode(0.0, {0.0, 1.0});
    for (int t = 0; t < 4; t++) CHECK(Eq(s.Sample(t), {(double)t, (double)t + 1}));
    CHECK_THROWS(s.AddNode(1.5, {0.0, 0.0}));
  }
  {  // ZeroOrder / Linear / Cubic (:120-163)
    TimeSpline z(2, kZeroSpline), l(2, kLinearSpline), c(2, kCubicSpline);
    for (TimeSpline* s : {&z, &l, &c}) { s->AddNode(1.0, {1.0, 2.0}); s->AddNode(2.0, {3.0, 4.0}); }
    CHECK(Eq(z.Sample(1.5), {1.0, 2.0}) && Eq(l.Sample(1.5), {2.0, 3.0}) && Eq(c.Sample(1.5), {2.0, 3.0}));
    c.Clear();
    c.AddNode(0.0, {1.0, 2.0}); c.AddNode(1.0, {1.0, 2.0}); c.AddNode(2.0, {3.0, 4.0}); c.AddNode(3.0, {3.0, 4.0});
    CHECK(Eq(c.Sample(1.5), {2.0, 3.0}));
    TimeSpline k(1, kCubicSpline);
    k.AddNode(-1.0, {1.0}); k.AddNode(0.0, {0.0}); k.AddNode(1.0, {1.0});
    for (double x = 0.0; x <= 1.0; x += 0.125) CHECK(k.Sample(x)[0] == -std::pow(x, 3) + 2 * std::pow(x, 2));
  }
  {  // ShiftTime (:165-185), ring loop (:233-257), Clear (:368-383), copies (:313-366), Dim0 (:385-398)
    TimeSpline s(2, kLinearSpline);
    for (int k = 1; k <= 4; k++) s.AddNode(k, {(double)k, (double)k + 1});
    CHECK(Eq(s.Sample(1.5), {1.5, 2.5}));
    s.ShiftTime(1.5);
    CHECK(s.Size() == 4 && Eq(s.Sample(1.5), {1.0, 2.0}) && Eq(s.Sample(2.0), {1.5, 2.5}));
    TimeSpline r(1);
    for (int k = 1; k <= 4; k++) r.AddNode(k, {(double)k});
    CHECK(r.DiscardBefore(3) == 2);
    r.AddNode(5.0, {5.0}); r.AddNode(6.0, {6.0});
    CHECK(r.DiscardBefore(6.0) == 3 && r.Size() == 1 && r.Sample(1.0)[0] == 6.0);
    TimeSpline c(2);
    c.AddNode(1.0, {1.0, 2.0});
    c.Clear();
    CHECK(c.Size() == 0 && Eq(c.Sample(0), {0.0, 0.0}));
    c.AddNode(1.0);
    CHECK(Eq(c.Sample(0), {0.0, 0.0}));  // AddNode resets to zero (:102-118)
    TimeSpline a(2, kLinearSpline);
    a.AddNode(1.0, {1.0, 2.0}); a.AddNode(2.0, {2.0, 3.0});
    a.DiscardBefore(2.0);
    a.AddNode(3.0, {3.0, 4.0}); a.AddNode(4.0, {4.0, 5.0}); a.AddNode(5.0, {5.0, 6.0});
    TimeSpline b(a);
    TimeSpline d(3);
    d = a;
    a.Clear();
    CHECK(b.Size() == 4 && Eq(b.Sample(2.5), {2.5, 3.5}) && Eq(d.Sample(1.5), {2.0, 3.0}) && d.Dim() == 2);
    TimeSpline e(0, kZeroSpline);
    e.AddNode(1.0); e.AddNode(2.0);
    CHECK(e.Size() == 2 && e.DiscardBefore(2.0) == 1 && e.Sample(1).empty());
  }
  TEST_MAIN_END();
}
